#!/usr/bin/env python3
"""Instruction classes of the main loop of a kernel of the SHIPPED library (llvm-objdump of the gfx950 code object
inside libmrgingham_amd.so): full-rate VALU, half-rate VALU (every v_pk_*, v_add3, v_perm, v_lshl*, v_mul_u32_u24, v_dot2,
... : tools/ubench/valu_rates.hip), LDS, SALU, memory.  The loop is the span of the longest backward branch.
usage: tools/isa_loop_classes.py [kernel-name-substring] [library.so]"""
import re, struct, subprocess, sys, tempfile, os, collections
kern = sys.argv[1] if len(sys.argv) > 1 else "chess_v1_pyr_kernel"
so = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mrgingham_amd", "libmrgingham_amd.so")
data = open(so, "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
text = None
for m in re.finditer(magic, data):
    i = m.start(); off = i + len(magic)
    (num,) = struct.unpack_from("<Q", data, off); off += 8
    for _ in range(num):
        o, s, tl = struct.unpack_from("<QQQ", data, off); off += 24
        triple = data[off:off + tl].decode(); off += tl
        if "gfx950" in triple and s > 0:
            with tempfile.NamedTemporaryFile(suffix=".o", delete=False) as f:
                f.write(data[i + o:i + o + s])
            dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
            os.unlink(f.name)
            mm = re.search(r"^[0-9a-f]+ <(\S*%s\S*)>:\n(.*?)(?=^\n[0-9a-f]+ <|\Z)" % re.escape(kern), dis, flags=re.S | re.M)
            if mm:
                text = mm
                break
    if text:
        break
assert text, "kernel not found"
ins = []
for l in text.group(2).split("\n"):
    mm = re.match(r"\s*(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", l)
    if mm:
        ins.append((int(mm.group(3), 16), mm.group(1), mm.group(2)))
best = None
for a, op, args in ins:
    if op.startswith("s_cbranch") or op == "s_branch":
        off = int(args.split()[-1])
        if off >= 32768:
            tgt = a + 4 + (off - 65536) * 4
            if best is None or a - tgt > best[1] - best[0]:
                best = (tgt, a)
lo, hi = best
FULL = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_lshrrev_b32", "v_ashrrev_i32",
        "v_cndmask_b32", "v_not_b32", "v_add_f32", "v_sub_f32", "v_mul_f32", "v_fma_f32", "v_max_u16", "v_max_i16", "v_add_u16", "v_bitop3_b32",
        "v_cmp_eq_u32", "v_cmp_ne_u32", "v_cmp_lt_u32", "v_cmp_gt_u32", "v_cmp_lt_i32", "v_cmp_gt_i32", "v_cmp_le_u32", "v_cmp_ge_u32",
        "v_cmp_ge_i32", "v_cmp_le_i32", "v_cmp_ne_u16", "v_readfirstlane_b32", "v_mbcnt_lo_u32_b32", "v_mbcnt_hi_u32_b32", "v_accvgpr_write_b32",
        "v_accvgpr_read_b32", "v_nop"}
cls = collections.Counter(); ops = collections.Counter()
for a, op, args in ins:
    if not (lo <= a <= hi):
        continue
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if op.startswith("v_"):
        half = base not in FULL or op.endswith(("_sdwa", "_dpp"))
        cls["VALU half rate" if half else "VALU full rate"] += 1
        ops[("h " if half else "f ") + base] += 1
    elif op.startswith("ds_"):
        cls["LDS"] += 1
    elif op.startswith(("buffer_", "global_", "flat_")):
        cls["memory"] += 1
    elif op.startswith("s_waitcnt"):
        cls["s_waitcnt"] += 1
    elif op.startswith("s_"):
        cls["SALU / control"] += 1
print(f"{text.group(1)[:60]}: main loop 0x{lo:x} .. 0x{hi:x} ({(hi - lo + 4)} bytes, {sum(cls.values())} instructions, every path counted once)")
for k, v in sorted(cls.items(), key=lambda x: -x[1]):
    print(f"  {k:18s} {v}")
print("  VALU opcodes:", ", ".join(f"{k} x{v}" for k, v in sorted(ops.items(), key=lambda x: -x[1])[:28]))
