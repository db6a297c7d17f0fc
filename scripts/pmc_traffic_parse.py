"""Parse the counter_collection CSVs of scripts/pmc_traffic_r02.sh / _r03.sh into gpurun_out/<round>_pmc_traffic.json
(round tag = first argument, default r03).
traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch: the counters are in KiB and gfx950's FETCH_SIZE reports
half of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM)."""
import csv, glob, json, os, statistics, sys
ROUND = sys.argv[1] if len(sys.argv) > 1 else 'r03'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = {'c5': 6 * 16 * 44, 'p4_1600': 6 * 40 * 100}
out = {'_how': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in SEPARATE passes (counters only) over `python3 bench.py --workload W '
               '--dtype D --steps 4 --warmup 2 --timed-only` (MI355X; scripts/pmc_traffic_' + ROUND + '.sh).  Counter unit KiB; '
               'traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch (gfx950 FETCH_SIZE counts half of a wide coalesced read). '
               'Cross- and self-attention launches of the fp32 kernels alternate in dispatch order (forward: self, cross; backward: '
               'cross, self), and so do those of the bf16 kernels in bf16 mode.  Medians over the launches of 6 steps.  '
               'dropout_mask_bytes: the packed keep-mask the training forward writes for the backward (inside WRITE_SIZE, outside '
               'algorithmic_min_bytes).'}
for wl, dt in (('c5', 'fp32'), ('p4_1600', 'bf16')):
    vals = {}
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        files = glob.glob(os.path.join(ROOT, 'gpurun_out', f'pmc_{wl}_{dt}_{c}', '**', '*counter_collection.csv'), recursive=True)
        if not files:
            continue
        rows = list(csv.DictReader(open(files[0])))
        per = {}
        for r in rows:
            if r.get('Counter_Name') != c:
                continue
            name = r['Kernel_Name']
            key = None
            if 'mha_fwd_bf16_kernel' in name: key = 'fwd16'
            elif 'mha_bwd_bf16_kernel' in name: key = 'bwd16'
            elif 'mha_fwd_kernel' in name: key = 'fwd'
            elif 'mha_bwd_kernel' in name or 'mha_bwd_sk_kernel' in name: key = 'bwd'
            elif 'coords3d_kernel' in name: key = 'coords3d'
            if dt == 'bf16' and key in ('fwd', 'bwd'):
                continue                      # (bf16 mode: no fp32 attention launches since the self-attention moved to the bf16 kernels)
            if key:
                per.setdefault(key, []).append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
        for k, lst in per.items():
            lst.sort()
            v = [x for _, x in lst]
            if k in ('fwd', 'fwd16'): v = v[1::2]        # self, cross, self, cross ...  (bf16 mode: both attentions run the bf16 kernels)
            if k in ('bwd', 'bwd16'): v = v[0::2]        # cross, self, ...
            vals.setdefault(k, {})[c] = statistics.median(v)
    Ltok = SHAPES[wl]
    ent = {}
    names = {'fwd': 'mha_fwd_cross', 'fwd16': 'mha_fwd_cross', 'bwd': 'mha_bwd_cross', 'bwd16': 'mha_bwd_cross', 'coords3d': 'coords3d'}
    for k, d in vals.items():
        if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
            e = {'dtype': dt, 'FETCH_SIZE_KiB': round(d['FETCH_SIZE'], 1), 'WRITE_SIZE_KiB': round(d['WRITE_SIZE'], 1),
                 'traffic_bytes': int((2 * d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024)}
            if k in ('fwd', 'fwd16'):
                eb = 4 if k == 'fwd' else 2
                e['algorithmic_min_bytes'] = 2 * Ltok * 256 * eb + 2 * 900 * 256 * 4
                e['dropout_mask_bytes'] = 900 * Ltok          # training forward also leaves 8 heads x 1 bit per (query, key)
            if k in ('bwd', 'bwd16'):
                eb = 4 if k == 'bwd' else 2
                e['algorithmic_min_bytes'] = 2 * Ltok * 256 * eb + 2 * Ltok * 256 * 4 + 4 * 900 * 256 * 4
            ent[names[k]] = e
    out[f'{wl}_{dt}'] = ent
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', ROUND + '_pmc_traffic.json'), 'w'), indent=1)
print(json.dumps(out, indent=1))
