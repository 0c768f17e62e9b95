#!/bin/bash
# Pastes tools/sessions/experiments/last_conv_experiments.inc into a scratch copy of csrc/edge_layers.hip (kernels in front of device_cu_count(), launchers behind it)
# and compiles it for gfx950: the archived experiment kernels still build against the library's helpers.  Nothing is written into the tree.
set -e
R=$(cd "$(dirname "$0")/../../.." && pwd); T=$(mktemp -d)
python3 - "$R" "$T" <<'PY'
import sys
R, T = sys.argv[1:3]
src = open(R + "/livespeechportraits_amd/csrc/edge_layers.hip").read()
inc = open(R + "/tools/sessions/experiments/last_conv_experiments.inc").read()
marker = "\nstatic int device_cu_count()\n"
assert marker in src
head, tail = src.split(marker, 1)
end = tail.index("\n}\n") + 3                      # the end of device_cu_count()
out = head + "\n" + inc.split("// ---- launchers")[0] + marker + tail[:end] + "\n#define LAST_CONV_EXPERIMENT_LAUNCHERS\n" + "// ---- launchers" + inc.split("// ---- launchers")[1] + tail[end:]
open(T + "/edge_layers_exp.hip", "w").write(out)
PY
cp "$R"/livespeechportraits_amd/csrc/*.h "$T"/; mkdir -p "$T/../include_dummy"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$R/livespeechportraits_amd/csrc" -I"$R/include" -c "$T/edge_layers_exp.hip" -o "$T/exp.o" -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|last_conv_ks|last_conv_sv|last_conv_vl" -A4 | grep -E "error|Function Name|VGPRs:|Scratch" | head -12
echo "compiled: $T/exp.o"; rm -rf "$T"
