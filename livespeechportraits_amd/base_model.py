"""Minimal inference-side counterpart of the reference's ``models/base_model.py``.

Keeps the conventions the render path relies on (reference file:line):
  :42      device = cuda:{gpu_ids[0]} if gpu_ids else cpu
  :87-97   setup(opt): inference => load_networks(opt.load_epoch)
  :108-113 eval()
  :161-176 save_networks: '<epoch>_<name>.pkl' = net.state_dict()
  :193-223 load_networks: '...pkl' epoch is a full path; CPU strips the 7-char 'module.'
           prefix; missing file at inference => ValueError
Training-side members (schedulers, optimizers, losses) are out of scope.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import torch


class BaseModel:
    def __init__(self, opt):
        self.opt = opt
        self.gpu_ids = list(opt.gpu_ids)
        self.isTrain = opt.isTrain
        self.device = torch.device("cuda:%d" % self.gpu_ids[0]) if self.gpu_ids else torch.device("cpu")
        self.save_dir = os.path.join(opt.checkpoints_dir, opt.name)
        self.model_names = []

    def setup(self, opt):
        if self.isTrain:
            raise NotImplementedError("training is out of scope of the HIP renderer")
        self.load_networks(opt.load_epoch)
        self.print_networks(getattr(opt, "verbose", False))

    def eval(self):
        for name in self.model_names:
            getattr(self, name).eval()

    def _path_for(self, epoch, name):
        epoch = str(epoch)
        if epoch.endswith("pkl"):
            return epoch
        return os.path.join(self.save_dir, "%s_%s.pkl" % (epoch, name))

    def save_networks(self, epoch, train_info=None):
        os.makedirs(self.save_dir, exist_ok=True)
        for name in self.model_names:
            torch.save(getattr(self, name).state_dict(), self._path_for(epoch, name))

    def load_networks(self, epoch):
        for name in self.model_names:
            path = self._path_for(epoch, name)
            net = getattr(self, name)
            if not os.path.exists(path):
                print("No model weight file:", path)
                if not self.isTrain:
                    raise ValueError("We are now in inference process, no pre-trained model found! "
                                     "Check the model checkpoint!")
                continue
            state = torch.load(path, map_location="cpu")
            want = set(net.state_dict().keys())
            fixed = OrderedDict()
            for k, v in state.items():
                # checkpoints are written from a DataParallel-wrapped net ('module.' prefix);
                # accept them on either wrapper, instead of the reference's blind 7-char strip
                if k not in want:
                    if k.startswith("module.") and k[7:] in want:
                        k = k[7:]
                    elif ("module." + k) in want:
                        k = "module." + k
                fixed[k] = v
            print("loading the model from %s" % path)
            missing = [k for k in want if k not in fixed and not k.endswith("num_batches_tracked")]
            if missing:
                # the reference loads with strict=False and would silently render garbage
                raise KeyError("checkpoint %s lacks %d generator tensors, e.g. %s" % (path, len(missing), sorted(missing)[:3]))
            net.load_state_dict(fixed, strict=False)

    def print_networks(self, verbose):
        for name in self.model_names:
            net = getattr(self, name)
            n = sum(p.numel() for p in net.parameters())
            if verbose:
                print(net)
            print("[Network %s] Total number of parameters : %.3f M" % (name, n / 1e6))
