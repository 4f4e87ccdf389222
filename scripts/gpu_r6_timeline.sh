#!/bin/bash
# Round 6: kernel timeline of the split encode leg with the product's two internal streams (rocprofv3 --kernel-trace: start / end of every
# dispatch) -- how long the persistent GEMMs run when they share the chip with the other micro-batch's kernels.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/timeline
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/timeline/two -o kt -- python scripts/encode_mode_leg.py split 1 16384 > gpurun_out/timeline/two.log 2>&1; echo "rc=$?"
ANCE_ENCODER_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/timeline/one -o kt -- python scripts/encode_mode_leg.py split 1 16384 > gpurun_out/timeline/one.log 2>&1; echo "rc=$?"
python - <<'PY'
import csv, glob, re, json
def short(n):
    m = re.search(r'(gemm256_split_stream_kernel<\d+>|gemm256_split_kernel<\d+>|attention_split_kernel|embed_split_kernel|head_gemm_kernel|head_ln_kernel|gather_cls_split_kernel|plan_kernel)', n)
    return m.group(1) if m else n[:40]
for v in ('two', 'one'):
    f = glob.glob('gpurun_out/timeline/%s/**/*kernel_trace.csv' % v, recursive=True)[0]
    rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r.get('Queue_Id', ''), int(r.get('Grid_Size', 0) or 0)) for r in csv.DictReader(open(f))]
    rows.sort()
    # keep the last encode call only: the dispatches after the largest gap in the second half
    t0 = rows[0][0]
    out = [dict(s=(a - t0) / 1e3, e=(b - t0) / 1e3, k=k, q=q, g=g) for a, b, k, q, g in rows]
    json.dump(out, open('gpurun_out/timeline/%s_timeline.json' % v, 'w'))
    import collections
    agg = collections.defaultdict(list)
    for r in out[len(out) // 2:]:
        agg[r['k']].append(r['e'] - r['s'])
    print(v, {k: (len(x), round(sum(x) / len(x), 1), round(min(x), 1), round(max(x), 1)) for k, x in agg.items() if len(x) > 5})
PY
find gpurun_out/timeline -name "*kernel_trace.csv" -delete
