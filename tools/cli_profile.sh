d=$(mktemp -d -p /dev/shm); python - "$d" <<'PY'
import sys, os, torch
sys.path.insert(0, os.getcwd())
from mrgingham_amd import synth
d = sys.argv[1]
fr = synth.board_batch(64, 4096, 3072, 10, 0, device="cuda").cpu().numpy()
for i in range(256):
    with open(os.path.join(d, f"f{i:03d}.pgm"), "wb") as f:
        f.write(b"P5\n4096 3072\n255\n"); f.write(fr[i % 64].tobytes())
PY
for j in 1 4 16; do echo "--jobs $j"; env MRGINGHAM_AMD_CLI_TIMING=1 mrgingham_amd/bin/mrgingham-amd-from-image --jobs $j "$d/f*.pgm" 2>&1 >/dev/null | grep -v amdgpu.ids | head -20; done
for j in 1; do env MRGINGHAM_AMD_CLI_TIMING=1 mrgingham_amd/bin/mrgingham-amd-from-image --noclahe --blur 0 --jobs $j "$d/f*.pgm" 2>&1 >/dev/null | grep -v amdgpu.ids | head -3; done
rm -rf "$d"
