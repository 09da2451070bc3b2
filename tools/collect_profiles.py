#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/<tag>_*) into the small tracked summaries under profiles/.

    python tools/collect_profiles.py r01b

Inputs (all produced on the GPU box, see DESIGN.md §6 for the exact commands):
    <tag>_trace, <tag>_nf4dq_ffn, <tag>_int8_4096   rocprofv3 --kernel-trace --stats  of bench.py per workload
    <tag>_fetch / _write / _tcc / _sq / _lds        rocprofv3 --kernel-trace --pmc ... one counter set per pass
"""
import collections, csv, glob, json, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01b"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def stats(name, cmd, out):
    files = sorted(glob.glob(os.path.join(src, f"{tag}_{name}", "*", "*kernel_stats.csv")), key=os.path.getmtime)
    if not files:
        return
    rows = list(csv.DictReader(open(files[-1])))   # newest run
    with open(os.path.join(dst, out), "w") as o:
        o.write(f"# rocprofv3 --kernel-trace --stats --output-format csv -- {cmd}   (MI355X; kernel names truncated to 110 chars)\n")
        o.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev\n")
        for r in rows:
            o.write('"%s",%s,%s,%s,%s,%s,%s,%s\n' % (r["Name"][:110].replace('"', "'"), r["Calls"], r["TotalDurationNs"],
                                                    r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]))


stats("trace", "python3 bench.py --no-cpu-baseline", "r01_bench_kernel_stats.csv")
stats("nf4dq_ffn", "python3 bench.py --workload nf4dq_ffn --no-cpu-baseline --steps 50", "r01_nf4dq_ffn_kernel_stats.csv")
stats("int8_4096", "python3 bench.py --workload int8_4096 --no-cpu-baseline --steps 50", "r01_int8_kernel_stats.csv")

res, dur = {}, []
for name in ("fetch", "write", "tcc", "sq", "lds"):
    files = sorted(glob.glob(os.path.join(src, f"{tag}_{name}", "*", "*counter_collection.csv")), key=os.path.getmtime)[-1:]
    if not files:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if "k_gemm256p" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if name == "sq" and r["Counter_Name"] == "SQ_WAVE_CYCLES":
                dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in agg.items():
        res[k] = {"mean": sum(v) / len(v), "dispatches": len(v)}
if "FETCH_SIZE" in res:
    fetch, write = res["FETCH_SIZE"]["mean"], res["WRITE_SIZE"]["mean"]
    out = {
        "kernel": "k_gemm256p<bf16, plain absmax> (mfma256), M=N=K=4096, NF4 bs64",
        "source": "rocprofv3 --kernel-trace --pmc <set> -- python3 bench.py --no-cpu-baseline --no-gemv --steps 5 --warmup 3; "
                  "separate passes: FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum | SQ_* | SQ_LDS_*",
        "gfx950_correction": "FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads: doubled (MI355X_MICROARCH.md, HBM)",
        "FETCH_SIZE_KB_raw": fetch, "WRITE_SIZE_KB": write,
        "fetch_bytes_per_launch": int(2 * fetch * 1024), "write_bytes_per_launch": int(write * 1024),
        "k_gemm256p_bytes_per_launch": int((2 * fetch + write) * 1024),
        "algorithmic_bytes_per_launch": 76546048,
        "l2_hit_rate": res["TCC_HIT_sum"]["mean"] / (res["TCC_HIT_sum"]["mean"] + res["TCC_MISS_sum"]["mean"]),
    }
    if dur and "SQ_WAVE_CYCLES" in res:
        d = sum(dur) / len(dur)
        wc = res["SQ_WAVE_CYCLES"]["mean"]
        out["profiled_duration_us"] = d / 1e3
        out["effective_clock_ghz"] = wc * 4 / 2048 / d          # SQ_WAVE_CYCLES counts quad-cycles summed over 2048 waves
        out["mfma_pipe_utilisation"] = res["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / 1024 / (wc * 4 / 2048)
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
            if k in res:
                out[k + "_per_wave_cycle"] = res[k]["mean"] / wc
    out["counters"] = res
    json.dump(out, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "counters"}, indent=1))
