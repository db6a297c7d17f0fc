#!/bin/bash
# MFMA / VALU / LDS utilisation counters of the four cross-attention kernels on the shapes bench.py times:
# rocprofv3 --pmc passes (counters only, never with a trace domain; program directly after "--") over the timed-only bench
# command for the headline (c5 fp32: mha_fwd_kernel, mha_bwd_sk_kernel) and for p4_1600 bf16 (mha_fwd_bf16_kernel,
# mha_bwd_bf16_kernel).  Cross-attention launches are told from self-attention ones by their position in the dispatch order
# (they alternate; see the parser).  Writes gpurun_out/r03_pmc_attention.txt.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r03_pmc_attention.txt
: > $out
for cfg in "c5 fp32" "p4_1600 bf16"; do
  set -- $cfg
  i=0
  for cset in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
              "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES" \
              "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
              "SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    d=/tmp/pmca_$1_$2_$i
    rm -rf $d
    (cd /tmp && rocprofv3 --pmc $cset -d $d -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload $1 --dtype $2 --steps 4 --warmup 2 --timed-only > $d.log 2>&1) || { echo "pass $1 $2 $i FAILED"; tail -5 $d.log; exit 1; }
    echo "pass $1 $2 $i done"
  done
  python3 - $1 $2 >> $out <<'PY'
import csv, sys, glob, collections
wl, dt = sys.argv[1], sys.argv[2]
vals = collections.defaultdict(lambda: collections.defaultdict(list))     # kernel -> counter -> values
grid = collections.defaultdict(int)
rows = []
for i in range(1, 5):
    for f in glob.glob(f'/tmp/pmca_{wl}_{dt}_{i}/**/*counter_collection.csv', recursive=True):
        rows += list(csv.DictReader(open(f)))
def short(n):
    for k in ('mha_fwd_bf16_kernel', 'mha_bwd_bf16_kernel', 'mha_fwd_kernel', 'mha_bwd_sk_kernel', 'mha_bwd_kernel'):
        if k in n:
            return k
    return None
# cross- and self-attention launches of a kernel alternate in dispatch order (forward: self, cross; backward: cross, self - in
# bf16 mode both attentions run the bf16 kernels, and the fp32 forward uses the same grid for both): per counter, order the
# launches by dispatch id and keep every second one
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = short(r['Kernel_Name'])
    if k:
        per[k][r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value']), int(r['Grid_Size'])))
for k, cs in per.items():
    first = 1 if 'fwd' in k else 0
    for c, lst in cs.items():
        lst.sort()
        sel = lst[first::2]
        vals[k][c] = [v for _, v, _ in sel]
        grid[k] = sel[0][2] if sel else 0
cyc_per_mfma = {'mha_fwd_kernel': 64, 'mha_bwd_kernel': 64, 'mha_bwd_sk_kernel': 64, 'mha_fwd_bf16_kernel': 32, 'mha_bwd_bf16_kernel': 32}
for k, cs in vals.items():
    if (dt == 'fp32') != ('bf16' not in k):
        continue
    print(f'## {k}, cross-attention launches of `bench.py --workload {wl} --dtype {dt} --timed-only` (grid {grid[k]} threads)')
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    for c in sorted(m):
        print(f'{c:32s} {m[c]:16.0f}  (mean of {len(cs[c])} launches)')
    if 'GRBM_GUI_ACTIVE' in m and 'SQ_VALU_MFMA_BUSY_CYCLES' in m:
        simd_cycles = 1024 * m['GRBM_GUI_ACTIVE'] / 8
        print(f'derived: MFMA pipe busy {m["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles:.3f} of SIMD-cycles '
              f'(= SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)); check: SQ_INSTS_MFMA x {cyc_per_mfma[k]} = '
              f'{m.get("SQ_INSTS_MFMA", 0) * cyc_per_mfma[k]:.0f}')
    if 'SQ_WAVE_CYCLES' in m:
        w = m['SQ_WAVE_CYCLES']
        print(f'derived: of a wave\'s cycles {m.get("SQ_WAIT_ANY", 0) / w:.2f} in s_waitcnt / barrier, {m.get("SQ_WAIT_INST_ANY", 0) / w:.2f} waiting to issue, '
              f'{m.get("SQ_ACTIVE_INST_ANY", 0) / w:.2f} issuing')
    if 'SQ_INSTS_VALU' in m and 'SQ_INSTS_MFMA' in m:
        print(f'derived: {m["SQ_INSTS_VALU"] / m["SQ_INSTS_MFMA"]:.2f} VALU and {m.get("SQ_INSTS_LDS", 0) / m["SQ_INSTS_MFMA"]:.2f} LDS instructions per MFMA; '
              f'LDS bank-conflict cycles {m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1):.3f} of LDS-active')
    print()
PY
done
cat $out | grep -E "^##|derived"
