"""CPU: the lane-level model of the matrix-core last conv (tools/lastconv_model.py) -- the index maps the kernel was written from.

last_conv_mfma (csrc/edge_layers.hip) places 16 source pixels x 4 output parities in the 16 blocks of v_mfma_f32_4x4x1_16b_f32, reads its A operand from an
XOR-swizzled LDS tile and its B operand from a padded weight table.  The model executes exactly those lane -> address maps in numpy; these tests keep it
equal to the direct sub-pixel convolution and every ds_read_b128 lane group on 16 distinct bank slots, so a change to either side shows up without a GPU.
(The MFMA operand layout the model assumes was read off the device: tools/probes/mfma4x4_probe.hip.)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import lastconv_model as M  # noqa: E402


def test_every_lane_group_of_the_operand_reads_is_conflict_free():
    M.check_banks()


def test_model_equals_the_direct_subpixel_convolution_on_one_tile_with_borders():
    # one 8 x 32 tile = the whole image: all four borders are out-of-range copies
    M.run(B=1, Hs=8, Ws=32, cout=3, seed=3)


def test_swizzle_is_a_permutation_of_a_pixels_eight_quads():
    for p in range(400):
        assert sorted(q ^ ((p >> 1) & 7) for q in range(8)) == list(range(8))


def test_stage_layout_puts_the_tile_interior_where_the_lanes_look():
    rng = np.random.default_rng(0)
    src = rng.standard_normal((1, 16, 64, 64)).astype(np.float32)
    lds = M.stage(src, 0, 8, 32, 1)                          # tile (1, 1), upper channel half
    for lane in (0, 17, 42, 63):
        for wave in range(4):
            for tau in range(4):
                for t in range(4):
                    p, slot = M.a_addr(lane, wave, tau, t, 5)
                    g, par, m = lane >> 4, (lane >> 2) & 3, lane & 3
                    y = 8 + 2 * wave + (tau >> 1) + (t >> 1) - 1 + (par >> 1)
                    x = 32 + 16 * (tau & 1) + 4 * g + m + (t & 1) - 1 + (par & 1)
                    want = src[0, y, x, 32 + 20: 32 + 24] if 0 <= y < 16 and 0 <= x < 64 else np.zeros(4, np.float32)
                    assert np.array_equal(lds[p, slot], want)
