#!/bin/bash
# Developer tool: register / spill / scratch figures of the frame kernels' bench instantiations, from the compiler's own
# resource remarks (-Rpass-analysis=kernel-resource-usage) and the static instruction mix of the device ISA (--save-temps).
# usage: tools/kernel_resources.sh [extra hipcc flags]   (run from anywhere; writes nothing into the tree)
cd "$(dirname "$0")/../tarl-simulator_amd/csrc"
T=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Rpass-analysis=kernel-resource-usage \
  --save-temps=obj "$@" -c fused.hip -o $T/fused.o 2> $T/remarks.txt
python3 - $T <<'PY'
import re, sys, glob
T = sys.argv[1]
want = {"_Z12k_fused_rowsILi4ELb1ELb0ELb1EE": "k_fused_rows<4,SIB,rollout,O32>", "_Z17k_fused_directionILi4ELb1ELb1ELb1EE": "k_fused_direction<4,SIB,CNT,O32>",
        "_Z15k_fused_insert2ILi8EE": "k_fused_insert2<8>", "_Z15k_fused_insert2ILi2EE": "k_fused_insert2<2>", "_Z14k_fused_insertill": "k_fused_insert"}
txt = open(T + "/remarks.txt").read()
blocks = re.split(r"Function Name: ", txt)[1:]
res = {}
for b in blocks:
    name = b.split()[0]
    for k, nice in want.items():
        if name.startswith(k):
            g = lambda pat, b=b: (re.search(pat + r": (\d+)", b) or [None, "?"])[1]
            res[nice] = dict(sgpr=g("TotalSGPRs"), vgpr=g("VGPRs"), sspill=g("SGPRs Spill"), vspill=g("VGPRs Spill"), scratch=g(r"ScratchSize \[bytes/lane\]"),
                             occ=g(r"Occupancy \[waves/SIMD\]"), lds=g(r"LDS Size \[bytes/block\]"))
asm = glob.glob(T + "/*gfx950*.s")
mix = {}
if asm:
    cur = None
    for line in open(asm[0]):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = next((nice for k, nice in want.items() if m.group(1).startswith(k)), None)
            if cur:
                mix[cur] = dict(valu=0, salu=0, vmem=0, smem=0, lds=0, lanespill=0)
            continue
        if cur is None:
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None
            continue
        op = line.strip().split(" ")[0].split("\t")[0]
        if op.startswith("v_readlane") or op.startswith("v_writelane"):
            mix[cur]["lanespill"] += 1
        if op.startswith("v_"):
            mix[cur]["valu"] += 1
        elif op.startswith("s_load") or op.startswith("s_buffer_load"):
            mix[cur]["smem"] += 1
        elif op.startswith("s_"):
            mix[cur]["salu"] += 1
        elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
            mix[cur]["vmem"] += 1
        elif op.startswith("ds_"):
            mix[cur]["lds"] += 1
print(f"{'kernel':36s} {'SGPR':>4s} {'VGPR':>4s} {'sSpill':>6s} {'vSpill':>6s} {'scratch':>7s} {'occ':>3s} {'LDS':>6s} | static: {'VALU':>5s} {'SALU':>5s} {'SMEM':>4s} {'VMEM':>4s} {'DS':>3s} {'v_read/writelane':>16s}")
for nice in want.values():
    if nice in res:
        r, m = res[nice], mix.get(nice, {})
        print(f"{nice:36s} {r['sgpr']:>4s} {r['vgpr']:>4s} {r['sspill']:>6s} {r['vspill']:>6s} {r['scratch']:>7s} {r['occ']:>3s} {r['lds']:>6s} | "
              f"        {m.get('valu', 0):5d} {m.get('salu', 0):5d} {m.get('smem', 0):4d} {m.get('vmem', 0):4d} {m.get('lds', 0):3d} {m.get('lanespill', 0):16d}")
PY
rm -rf $T
