"""Aggregate rocprofv3 counter_collection CSVs (one counter per run) per kernel name, and the
kernel-trace stats CSVs, from the directory tree scripts/gpu_pmc.sh writes."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True)):
    print("== " + f)
    with open(f) as fh:
        for i, row in enumerate(csv.reader(fh)):
            if i < 12:
                print("   " + " | ".join(x[:70] for x in row))
for d in sorted(glob.glob(os.path.join(root, "*_*"))):
    base = os.path.basename(d)
    counter = base.rsplit("_", 1)[0]
    if counter not in ("FETCH_SIZE", "WRITE_SIZE"):
        continue
    agg = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = row.get("Kernel_Name", "?").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:80]
                agg[name][0] += float(row.get("Counter_Value", 0) or 0)
                agg[name][1] += 1
    print("== %s  (sum over dispatches / per dispatch; rocprofv3 reports KB; on gfx950 FETCH_SIZE counts wide "
          "streaming reads at half their bytes -- MI355X_MICROARCH.md HBM section)" % base)
    for name, (tot, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:12]:
        print("   %-82s dispatches %6d  total %16.1f  per-dispatch %14.2f" % (name, n, tot, tot / max(n, 1)))

# ---- bytes per launch of the kernels bench.py reports a roofline for -> JSON (profiles/pmc_traffic.json)
if len(sys.argv) > 2:
    import json

    def per_dispatch(leg, counter, needle):
        tot, n = 0.0, 0
        for f in glob.glob(os.path.join(root, "%s_%s" % (counter, leg), "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    if row.get("Counter_Name") == counter and needle in row.get("Kernel_Name", ""):
                        tot += float(row.get("Counter_Value", 0) or 0)
                        n += 1
        return (tot / n * 1024.0) if n else None  # rocprofv3 reports KB

    def cycles(leg, needle):
        """(shader cycles per XCD, MFMA-pipe busy fraction, effective clock GHz) per dispatch, from the
        CYCLES pass: GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs;
        the clock is cycles / the dispatch's own duration in the same run's kernel trace."""
        d = os.path.join(root, "CYCLES_%s" % leg)
        gui, busy, dur = [], [], []
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    if needle not in row.get("Kernel_Name", ""):
                        continue
                    if row.get("Counter_Name") == "GRBM_GUI_ACTIVE":
                        gui.append(float(row["Counter_Value"]) / 8.0)
                    elif row.get("Counter_Name") == "SQ_VALU_MFMA_BUSY_CYCLES":
                        busy.append(float(row["Counter_Value"]) / 1024.0)
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    if needle in row.get("Kernel_Name", ""):
                        dur.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        if not gui or not dur:
            return None
        g, t = sum(gui) / len(gui), sum(dur) / len(dur)
        return {"cycles_per_launch": g, "mfma_busy_frac": (sum(busy) / len(busy) / g) if busy else None,
                "clock_ghz": g / t, "ns_per_launch_under_pmc": t}

    def l2_hit(leg, needle):
        hit = miss = 0.0
        for f in glob.glob(os.path.join(root, "L2_%s" % leg, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    if needle not in row.get("Kernel_Name", ""):
                        continue
                    if row.get("Counter_Name") == "TCC_HIT_sum":
                        hit += float(row["Counter_Value"])
                    elif row.get("Counter_Name") == "TCC_MISS_sum":
                        miss += float(row["Counter_Value"])
        return hit / (hit + miss) if hit + miss > 0 else None

    def trace_avg_ns(leg, needle):
        """average dispatch duration in the plain kernel-trace run (no counters)"""
        dur = []
        for f in glob.glob(os.path.join(root, "kt_%s" % leg, "**", "*kernel_trace.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    if needle in row.get("Kernel_Name", ""):
                        dur.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        return (sum(dur) / len(dur), len(dur)) if dur else (None, 0)

    def step_traffic():
        """FETCH x 2 + WRITE bytes of EVERY dispatch of one encode step (the *_encodeall passes: 4,096-passage block, one
        warm-up step and one timed step: the second half of the dispatches), and the untraced time of that step."""
        tot = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            rows = []
            for f in glob.glob(os.path.join(root, "%s_encodeall" % counter, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row.get("Counter_Name") == counter:
                            rows.append((int(row.get("Dispatch_Id", 0) or 0), float(row.get("Counter_Value", 0) or 0), row.get("Kernel_Name", "")))
            if not rows:
                return None
            rows.sort()
            enc = [r for r in rows if any(k in r[2] for k in ("gemm256", "attention", "embed", "head", "plan_kernel", "pack_kernel", "gather_cls"))]
            # bench.py --steps 1 --warmup 1 runs the block four times (warm-up + timed step, then the same on the single-stream
            # roofline handle): the last quarter of the encoder dispatches is one step
            half = enc[len(enc) - len(enc) // 4:]
            tot[counter] = sum(v for _, v, _ in half) * 1024.0
            tot[counter + "_dispatches"] = len(half)
        try:
            with open(os.path.join(root, "encode_all_plain.json")) as fh:
                line = json.loads(fh.read().strip().splitlines()[-1])
            ms, pps = line["ms_per_step"], line["value"]
        except Exception:
            return None
        b = 2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]
        return {"block_passages": 4096, "dispatches": tot["FETCH_SIZE_dispatches"], "fetch_bytes_x2": 2.0 * tot["FETCH_SIZE"],
                "write_bytes": tot["WRITE_SIZE"], "hbm_bytes_per_step": b, "ms_per_step_untraced_single_stream": ms,
                "hbm_gbs_rocprof": b / (ms * 1e-3) / 1e9, "bytes_per_passage": b / 4096.0, "passages_per_sec": pps,
                "round": os.environ.get("ANCE_ROUND", "r06"), "encoder_precision": line.get("encoder_precision"),
                "note": "sum over every encoder dispatch of one step (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                        "FETCH_SIZE doubled per MI355X_MICROARCH.md) / the untraced time of the same step; the algorithmic minimum is "
                        "4 + 4 L + 3072 bytes per passage -- the rest is activation round trips between the kernels of a layer"}

    out = {}
    st = step_traffic()
    if st:
        # the step that was counted is the bench's headline mode: split since round 5 (fp16 before)
        out.setdefault("encode_split" if st.get("encoder_precision") == "split" else "encode", {})["whole_step"] = st
    # template arguments of gemm256_f16_desc_kernel: gemm_f16.h (4 = RESLN: attention.output.dense and output.dense, 5 = QK_F,
    # 6 = GELU_F: intermediate.dense, 7 = VT_F)
    for leg, cat, needle in (("encode", "gemm_ffn1", "gemm256_f16_desc_kernel<6"), ("encode", "gemm_qk", "gemm256_f16_desc_kernel<5"),
                             ("encode", "gemm_res", "gemm256_f16_desc_kernel<4"), ("encode", "gemm_vt", "gemm256_f16_desc_kernel<7"),
                             ("encode", "attention", "attention_kernel"),
                             # (round 6: QKV and FFN1 run the persistent streaming kernel)
                             ("encode_split", "gemm_ffn1", "gemm256_split_stream_kernel<9"), ("encode_split", "gemm_qk", "gemm256_split_stream_kernel<8"),
                             ("encode_split", "gemm_ffn1_per_tile", "gemm256_split_kernel<9"), ("encode_split", "gemm_qk_per_tile", "gemm256_split_kernel<8"),
                             ("encode_split", "gemm_res", "gemm256_split_kernel<10"), ("encode_split", "attention", "attention_split_kernel"),
                             ("search", "ip_topk_fast", "ip_topk_fast_kernel<false, false>"), ("search", "ip_topk_rescore", "rescore_kernel"),
                             ("search", "ip_topk_scan", "ip_topk_scan_kernel")):
        fe, wr = per_dispatch(leg, "FETCH_SIZE", needle), per_dispatch(leg, "WRITE_SIZE", needle)
        if fe is None or wr is None:
            continue
        avg_ns, n_disp = trace_avg_ns(leg, needle)
        out.setdefault(leg, {})[cat] = {"cycles": cycles(leg, needle), "l2_hit_rate": l2_hit(leg, needle),
                                        "kernel_trace_avg_ns": avg_ns, "kernel_trace_dispatches": n_disp,
                                        "hbm_bytes_per_launch": 2.0 * fe + wr, "fetch_bytes_x2": 2.0 * fe, "write_bytes": wr,
                                        "round": os.environ.get("ANCE_ROUND", "r06"),
                                        "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT,MISS / cycles (separate passes) on the bench.py leg "
                                                "itself (scripts/gpu_pmc.sh); FETCH_SIZE doubled per MI355X_MICROARCH.md (wide reads on gfx950)"}
    with open(sys.argv[2], "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))
