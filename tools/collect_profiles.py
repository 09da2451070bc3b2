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


rnd = tag[:3]   # "r02x" -> files named r02_*
stats("trace", "python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-empirical", f"{rnd}_bench_kernel_stats.csv")
stats("nf4dq_ffn", "python3 bench.py --workload nf4dq_ffn --no-cpu-baseline --steps 50", f"{rnd}_nf4dq_ffn_kernel_stats.csv")
stats("int8_4096", "python3 bench.py --workload int8_4096 --no-cpu-baseline --steps 50", f"{rnd}_int8_kernel_stats.csv")
stats("nf4_m1", "python3 bench.py --workload nf4_m1 --no-cpu-baseline --steps 20", f"{rnd}_gemv_kernel_stats.csv")
for fn in ("bench_driver", "bench"):
    p = os.path.join(src, f"{tag}_{fn}.json")
    if os.path.exists(p):
        lines = [l for l in open(p) if l.startswith("{")]
        if lines:
            open(os.path.join(dst, f"{rnd}_{fn}_line.json"), "w").write(lines[-1])


def kernel_counter_mean(dirname, kernel_substr, counter):
    files = sorted(glob.glob(os.path.join(src, dirname, "*", "*counter_collection.csv")), key=os.path.getmtime)[-1:]
    if not files:
        return None
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(files[0]))
         if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return (sum(v) / len(v), len(v)) if v else None

res, dur = {}, []
for name in ("fetch", "write", "tcc", "sq", "lds"):
    files = sorted(glob.glob(os.path.join(src, f"{tag}_{name}", "*", "*counter_collection.csv")), key=os.path.getmtime)[-1:]
    if not files:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if "k_gemm_dense" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if name == "sq" and r["Counter_Name"] == "SQ_WAVE_CYCLES":
                dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in agg.items():
        res[k] = {"mean": sum(v) / len(v), "dispatches": len(v)}
if "FETCH_SIZE" in res:
    fetch, write = res["FETCH_SIZE"]["mean"], res["WRITE_SIZE"]["mean"]
    out = {
        "kernel": "k_gemm_dense<bf16> (dequant+dense), M=N=K=4096, weight dequantised once from NF4 bs64",
        "source": "rocprofv3 --kernel-trace --pmc <set> -- python3 bench.py --no-cpu-baseline --no-gemv --steps 5 --warmup 3; "
                  "separate passes: FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum | SQ_* | SQ_LDS_*",
        "gfx950_correction": "FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads: doubled (MI355X_MICROARCH.md, HBM)",
        "FETCH_SIZE_KB_raw": fetch, "WRITE_SIZE_KB": write,
        "fetch_bytes_per_launch": int(2 * fetch * 1024), "write_bytes_per_launch": int(write * 1024),
        "k_gemm_dense_bytes_per_launch": int((2 * fetch + write) * 1024),
        "algorithmic_bytes_per_launch": 3 * 4096 * 4096 * 2,
        "l2_hit_rate": res["TCC_HIT_sum"]["mean"] / (res["TCC_HIT_sum"]["mean"] + res["TCC_MISS_sum"]["mean"]),
    }
    if dur and "SQ_WAVE_CYCLES" in res:
        d = sum(dur) / len(dur)
        wc = res["SQ_WAVE_CYCLES"]["mean"]
        out["profiled_duration_us"] = d / 1e3
        out["effective_clock_ghz"] = wc * 4 / 1024 / d          # SQ_WAVE_CYCLES counts quad-cycles summed over 1024 waves
        out["mfma_pipe_utilisation"] = res["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / 1024 / (wc * 4 / 1024)
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
            if k in res:
                out[k + "_per_wave_cycle"] = res[k]["mean"] / wc
    out["counters"] = res
    # HBM traffic of the GEMV (M = 1, 64 rotating layers) and of the int8 GEMM, same counters and correction
    for wl, sub, key, alg in (("nf4_m1", "k_gemv4", "k_gemv4_bytes_per_launch", 9453568),
                              ("int8_4096", "k_gemm_i8_inplace", "k_gemm_i8_bytes_per_launch", 2 * 4096 * 4096 + 4096 * 4096 * 2 + 2 * 4096 * 4),   # matmul_int8's single launch (round 3: B read in place)
                              ("nf4dq_ffn", "k_gemm_dense", "nf4dq_ffn_gemm_dense_bytes_per_launch", (4096 * 4096 + 11008 * 4096 + 4096 * 11008) * 2),
                              # OutlierAwareLinear's GEMM: int8 A and W, 16-bit out, scales, 16 outlier columns (compact activations, weights), bias
                              ("outlier", "k_gemm_dense", "k_gemm_i8_outlier_bytes_per_launch",
                               2 * 4096 * 4096 + 4096 * 4096 * 2 + 2 * 4096 * 4 + 2 * 4096 * 16 * 2 + 4096 * 2)):
        f = kernel_counter_mean(f"{tag}_{wl}_fetch", sub, "FETCH_SIZE")
        w = kernel_counter_mean(f"{tag}_{wl}_write", sub, "WRITE_SIZE")
        if f and w:
            out[key] = int((2 * f[0] + w[0]) * 1024)
            out[key.replace("_bytes_per_launch", "_detail")] = {"FETCH_SIZE_KB_raw": f[0], "WRITE_SIZE_KB": w[0], "dispatches": f[1],
                                                                "algorithmic_bytes_per_launch": alg}
    # bytes of a whole STEP (every launch of one matmul call): the line's roofline.traffic.  (FETCH_SIZE doubled per launch kind.)
    def step_bytes(dirtag, subs):
        tot = 0
        for sub in subs:
            f = kernel_counter_mean(f"{dirtag}_fetch", sub, "FETCH_SIZE")
            w = kernel_counter_mean(f"{dirtag}_write", sub, "WRITE_SIZE")
            if not (f and w):
                return None
            tot += int((2 * f[0] + w[0]) * 1024)
        return tot
    for key, dirtag, subs, alg in (("nf4_m4096_step_bytes", tag, ("k_dequantize_4bit", "k_gemm_dense"), 76546048),
                                   ("nf4dq_ffn_step_bytes", f"{tag}_nf4dq_ffn", ("k_dequantize_4bit", "k_gemm_dense"), 146991872),
                                   ("int8_4096_step_bytes", f"{tag}_int8_4096", ("k_gemm_i8_inplace",), 67141632)):
        v = step_bytes(dirtag, subs)
        if v is not None:
            out[key] = v
            out[key + "_algorithmic"] = alg
    # where these counters come from: bench.py repeats it next to every `traffic` figure so that a stale file is visible in the line
    import datetime, subprocess
    try:
        commit = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
        dirty = bool(subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "mps_bitsandbytes_amd", "bench.py"], capture_output=True, text=True).stdout.strip())
    except Exception:
        commit, dirty = "unknown", False
    mt = max((os.path.getmtime(f) for f in glob.glob(os.path.join(src, f"{tag}_fetch", "*", "*counter_collection.csv"))), default=None)
    out["_source"] = {"tag": tag, "commit": commit + ("+uncommitted" if dirty else ""),
                      "date": datetime.datetime.utcfromtimestamp(mt).strftime("%Y-%m-%d %H:%M UTC") if mt else "unknown",
                      "recipe": "tools/profile_round.sh " + tag + " on an MI355X box, condensed by tools/collect_profiles.py"}
    json.dump(out, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "counters"}, indent=1))
