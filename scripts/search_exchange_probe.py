"""NOTE: measures the two-phase search API of commit 3758d29 (reverted: see DESIGN.md section 9); check that commit out to re-run."""
"""What the threshold exchange between shards buys at the shard size of an 8-GPU job, measured on ONE GPU: the 8,841,823-row
corpus of the bench is generated shard by shard (1,105,228 rows each), every shard is scanned (ance_ip_topk_scan) for the same
32,768 queries -- exactly the bounds seven other ranks would send --, and shard 0's finish phase is timed with their maximum
against its stand-alone search (ance_ip_topk_indexed)."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from ance_amd import _lib  # noqa: E402
from ance_amd.index import FlatIPIndex  # noqa: E402

G, n_total, nq, k = 8, 8841823, 32768, 200
per = (n_total + G - 1) // G
gq = torch.Generator(device="cuda").manual_seed(99)
q = torch.nn.functional.layer_norm(torch.randn((nq, 768), generator=gq, device="cuda"), (768,))
lb = None
first = None
for r in range(G):
    g = torch.Generator(device="cuda").manual_seed(4321 + r)
    n = min(per, n_total - r * per)
    x = torch.nn.functional.layer_norm(torch.randn((n, 768), generator=g, device="cuda"), (768,))
    idx = FlatIPIndex(768, row_base=r * per)
    idx.add(x)
    l = idx.scan_device(q, k)
    lb = l if lb is None else torch.maximum(lb, l)
    if r == 0:
        first = (idx, x)
    else:
        del idx, x
idx, x = first


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / reps


res = {}
res["standalone_search_ms"] = timed(lambda: idx.search_device(q, k))
res["scan_ms"] = timed(lambda: idx.scan_device(q, k))


def pair(bound):
    idx.scan_device(q, k)
    return idx.finish_device(q, k, bound)


res["scan_plus_finish_with_exchange_ms"] = timed(lambda: pair(lb))
res["scan_plus_finish_without_bounds_ms"] = timed(lambda: pair(None))
_lib.profile_enable(True)
D, I = pair(lb)
torch.cuda.synchronize()
prof = _lib.profile_read()
_lib.profile_enable(False)
res["with_exchange_kernels_ms"] = {c: v["ms"] for c, v in prof.items() if v["count"]}
_lib.profile_enable(True)
D0, I0 = idx.search_device(q, k)
torch.cuda.synchronize()
prof = _lib.profile_read()
_lib.profile_enable(False)
res["standalone_kernels_ms"] = {c: v["ms"] for c, v in prof.items() if v["count"]}
res["entries_kept_per_query_with_exchange"] = float((I >= 0).sum().item()) / nq
res["rows_per_shard"], res["queries"], res["k"], res["shards"] = per, nq, k, G
# the kept entries are a subset of the stand-alone list, in the same order
m = I >= 0
res["kept_is_prefix_of_standalone"] = bool(torch.equal(torch.where(m, I, I0), I0))
print(json.dumps(res, indent=1))
json.dump(res, open("gpurun_out/search_exchange_probe.json", "w"), indent=1)
