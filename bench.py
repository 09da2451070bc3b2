#!/usr/bin/env python3
"""
bench.py — throughput of the NF4 dequant + matmul hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: N child ranks are started)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Timing: a disclosed time-based pre-warm ("prewarm_ms": the clock needs a few hundred ms of load to settle),
W untimed warm-up steps, then R = 5 repetitions of the K-step loop, each bracketed by barrier + synchronize on both
sides and reduced with MAX over ranks; `value` and `ms_per_step` come from the MEDIAN repetition, and the kernel's
launch duration (`roofline.kernel_us`) from HIP events recorded inside that same repetition on the launching stream.

One "step" = one pass of the hot path over one batch: Linear4bit-style matmul_4bit on an NF4 weight
4096x4096 (bf16-origin, blocksize 64), M = 4096 rows per GPU, bf16 activations, inputs resident in HBM.  At this M the
library dequantises the weight ONCE into a transient scratch and runs a dense MFMA GEMM (two launches per step, as the
reference does above M = 512); the line says which launches served the step (config.kernel, roofline.launches).  With N > 1 the weight is replicated, every rank owns 4096 rows of a global
batch of 4096*N (BASELINE configs[4] at N = 8: M = 32768) and the step ends with an RCCL
all-gather of the output shards over xGMI (weak scaling).

Rank 0 prints ONE JSON line: metric/value per the driver contract plus
  "roofline"     the OPERATION (the whole step, every launch of one matmul call; MFMA-bound): algorithmic flops / step duration
                 measured with HIP events on the launching stream; `dominant_kernel` carries the largest launch alone
  "cpu_baseline" the CPU oracle (a port of the reference's CPU path) timed on this box's host
                 cores on a bounded row-sample of the same workload
  "gemv"         the M = 1 decode shape of the metric (HBM-bound), rotating over 64 distinct
                 layers (605 MB > 256 MB Infinity Cache), with its own roofline fraction.
  "secondary"    (default workload, N = 1) the other BASELINE GPU configs as complete lines of their own: nf4dq_ffn (configs[2]),
                 int8_4096 (configs[3]), nf4_m1 (configs[1] stand-alone), each with roofline, traffic and cpu_baseline
  "no_prewarm"   the same K steps timed before the disclosed time-based pre-warm (cold clock)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "int8": 5000.0}   # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--no-empirical", action="store_true", help="skip the vendor-BLAS / copy-bandwidth context figures")
    ap.add_argument("--warmup", type=int, default=100)  # ~13 ms: the GPU clock needs a few ms of load to ramp
    ap.add_argument("--workload", default="nf4_m4096",
                    choices=["nf4_m4096", "nf4dq_ffn", "int8_4096", "nf4_m1", "embed4", "embed8", "outlier", "w8a16", "fp8"],
                    help="nf4_m4096 = the BASELINE metric (default); nf4dq_ffn / int8_4096 / nf4_m1 = BASELINE configs 3, 4, 2; "
                         "embed4 / embed8 / outlier = the SURVEY 8f rank-3 rows; w8a16 / fp8 = Linear8bit / LinearFP8 forward (single GPU)")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the output all-gather (GEMM-only scaling)")
    ap.add_argument("--sync-gather", action="store_true", help="N>1: blocking all-gather after every GEMM (no overlap); "
                    "with --no-gather and the default these are the three curves of SURVEY 8e")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gemv", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="default workload at N = 1: skip the secondary lines (BASELINE configs 2, 3 and stand-alone 1)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--prewarm-ms", type=float, default=400.0, help="untimed, time-based load before the warm-up steps")
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the K-step loop; the median one is reported")
    ap.add_argument("--chunks", type=int, default=2, help="N>1: row chunks of the `chunked` curve (GEMM chunk i+1 under the gather of chunk i)")
    ap.add_argument("--verify", action="store_true", help="N>1: check gathered output == unsharded result on rank 0")
    return ap.parse_args()


def probe_lib():
    """tools/libmbnb_probe.so (bare MFMA loop for the `roofline.empirical` context figures; NOT part of the product library),
    or None when it has not been built."""
    import ctypes
    path = os.path.join(ROOT, "tools", "libmbnb_probe.so")
    if not os.path.exists(path):
        return None
    lib = ctypes.CDLL(path)
    lib.mbnb_probe_mfma.restype = ctypes.c_int64
    lib.mbnb_probe_mfma.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    return lib


def bare_mfma_rate(kind):
    """Sustained rate (T op/s) of a bare MFMA loop on every SIMD of this box (kind 0: bf16 32x32x16, 1: i8 32x32x32), or None."""
    lib = probe_lib()
    if lib is None:
        return None
    sink = torch.zeros(1, dtype=torch.float32, device="cuda")
    stp = torch.cuda.current_stream().cuda_stream
    if lib.mbnb_probe_mfma(kind, 2000, sink.data_ptr(), stp) <= 0:
        return None
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n_mfma = 0
    for _ in range(10):
        n_mfma += int(lib.mbnb_probe_mfma(kind, 20000, sink.data_ptr(), stp))
    e1.record()
    e1.synchronize()
    per = 32768.0 if kind == 0 else 65536.0     # 2*32*32*16 / 2*32*32*32 operations per MFMA
    return n_mfma * per / (e0.elapsed_time(e1) * 1e-3) / 1e12 if n_mfma > 0 else None


def event_time_ms(fn, steps):
    """Average duration of fn() over `steps` back-to-back launches, HIP events on the current stream."""
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(steps):
        fn()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / steps


def _outlier_traffic():
    """HBM bytes per launch of the OutlierAwareLinear GEMM from the committed PMC passes (tools/profile_round.sh), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f).get("k_gemm_i8_outlier_bytes_per_launch")
    except (OSError, ValueError):
        return None


def bench_nn(args, wl, dev, bnb, synthetic):
    """SURVEY 8f rank-3 rows on one GPU: 4-bit / 8-bit embedding lookups (HBM-bound gathers over a 32000 x 4096 table,
    8192 looked-up rows per step) and OutlierAwareLinear.forward (4096^3, 16 outlier columns, int8 MFMA)."""
    import numpy as np
    import oracle
    out = {"n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "data": "synthetic"}
    if wl in ("embed4", "embed8"):
        num, dim, tokens, dt = 32000, 4096, 8192, torch.bfloat16
        W = synthetic.normal((num, dim), dt, seed=1234, std=0.5).to(dev)
        idx_cpu = torch.from_numpy((synthetic.uniform_u64(tokens, seed=4321) % np.uint64(num)).astype(np.int64))
        idx = idx_cpu.to(dev)
        if wl == "embed4":
            packed, st = bnb.quantize_nf4(W, blocksize=64)
            packed, absmax = packed.view(num, dim // 2), st.absmax.view(num, -1)
            step = lambda: bnb.embedding_4bit(idx, packed, absmax, dim, 64, "nf4", None, dt)
            bytes_per_row = dim // 2 + (dim // 64) * 4 + dim * 2 + 8
            cpu = lambda: oracle.embedding_4bit(idx_cpu, packed.cpu(), absmax.cpu(), dim, 64, "nf4", None, dt)
            label = "Embedding4bit (NF4 bs64)"
        else:
            q, sc = bnb.quantize_rowwise(W)
            step = lambda: bnb.embedding_8bit(idx, q, sc, None, dt)
            bytes_per_row = dim + 4 + dim * 2 + 8
            qc, scc = q.cpu(), sc.cpu()
            cpu = lambda: oracle.embedding_8bit(idx_cpu, qc, scc, None, dt)
            label = "Embedding8bit"
        del W
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        kern_ms = event_time_ms(step, args.steps)
        nbytes = bytes_per_row * tokens
        gbs = nbytes / (kern_ms * 1e-3) / 1e9
        out.update({"metric": f"effective GB/s, {label} lookup of {tokens} rows from a {num} x {dim} table -> bf16",
                    "value": round(nbytes / (elapsed / args.steps) / 1e9, 1), "unit": "GB/s",
                    "ms_per_step": round(elapsed / args.steps * 1e3, 5), "dtype": "u8" if wl == "embed4" else "int8",
                    "config": {"workload": f"{label}.forward, vocabulary {num}, embedding_dim {dim}, {tokens} indices per step",
                               "algorithmic_bytes_per_row": bytes_per_row},
                    "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None, "kernel_us": round(kern_ms * 1e3, 2)}})
        if not args.no_cpu_baseline:
            cpu()
            t0, n = time.perf_counter(), 0
            while time.perf_counter() - t0 < min(args.cpu_seconds, 5.0):
                cpu()
                n += 1
            dtc = (time.perf_counter() - t0) / n
            out["cpu_baseline"] = {"value": round(nbytes / dtc / 1e9, 2), "unit": "GB/s", "cores": oracle.num_threads(),
                                   "kind": "port", "sample": f"{n} x the full step on the host (oracle, OpenMP)"}
        return out
    if wl in ("w8a16", "fp8"):
        # Linear8bit / LinearFP8 forward at the metric shape: 16-bit activations x 8-bit weights decoded in the GEMM
        M = N = K = 4096
        dt = torch.bfloat16
        lin = torch.nn.Linear(K, N, bias=False)
        lin.weight.data.copy_(synthetic.normal((N, K), torch.float32, seed=1234, std=0.02))
        layer = (bnb.Linear8bit if wl == "w8a16" else bnb.LinearFP8).from_linear(lin.to(dt).to(dev))
        x = synthetic.normal((M, K), dt, seed=4321).to(dev)
        step = lambda: layer(x)
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        kern_ms = event_time_ms(step, args.steps)
        flops = 2.0 * M * N * K
        tf = flops / (kern_ms * 1e-3) / 1e12
        name = "Linear8bit (row-wise INT8 weights)" if wl == "w8a16" else "LinearFP8 (E4M3 weights, reference format)"
        out.update({"metric": f"effective bf16 TFLOPS, {name} forward 4096x4096 @ M=4096", "value": round(flops / (elapsed / args.steps) / 1e12, 1),
                    "unit": "TFLOP/s", "ms_per_step": round(elapsed / args.steps * 1e3, 5), "dtype": "bf16",
                    "config": {"workload": f"{name}.forward, bf16 activations, weights dequantised once into the workspace + k_gemm_dense at this M (fused W8A16 kernels below 512 rows)", "M": M, "N": N, "K": K,
                               "kernel": __import__("mps_bitsandbytes_amd")._native.last_kernel()},
                    "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": PEAK_TFLOPS["bf16"], "unit": "TFLOP/s",
                                 "frac": round(tf / PEAK_TFLOPS["bf16"], 4), "traffic": None, "kernel_us": round(kern_ms * 1e3, 2)}})
        if not args.no_cpu_baseline:
            rows = 256
            xs = x[:rows].cpu()
            if wl == "w8a16":
                q, sc = layer.weight_int8.cpu(), layer.weight_scales.cpu()
                t0 = time.perf_counter(); oracle.linear_int8(xs, q, sc); dtc = time.perf_counter() - t0
            else:
                q, sc = layer.weight_fp8.cpu(), layer.weight_scales.cpu()
                t0 = time.perf_counter(); oracle.linear_fp8(xs, q, sc); dtc = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": round(2.0 * rows * N * K / dtc / 1e12, 4), "unit": "TFLOP/s", "cores": oracle.num_threads(),
                                   "kind": "port", "sample": f"{rows} of the {M} rows, one pass (oracle, OpenMP)"}
        return out
    M = N = K = 4096
    dt, n_out = torch.float16, 16
    W = synthetic.normal((N, K), torch.float32, seed=1234, std=0.05)
    oidx = torch.arange(100, 100 + 250 * n_out, 250)
    W[:, oidx] *= 30.0
    lin = torch.nn.Linear(K, N, bias=True)
    lin.weight.data.copy_(W)
    oa = bnb.OutlierAwareLinear.from_linear(lin.to(dt).to(dev))
    assert int(oa.outlier_indices.numel()) == n_out
    x = synthetic.normal((M, K), dt, seed=4321).to(dev)
    step = lambda: oa(x)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_ms = event_time_ms(step, args.steps)
    ops = 2.0 * M * N * K
    tops = ops / (kern_ms * 1e-3) / 1e12
    out.update({"metric": "effective TOP/s, OutlierAwareLinear.forward 4096x4096 @ M=4096 (row-wise int8 quantise + int8 MFMA + 16 outlier columns + bias)",
                "value": round(ops / (elapsed / args.steps) / 1e12, 1), "unit": "TOP/s", "ms_per_step": round(elapsed / args.steps * 1e3, 5),
                "dtype": "int8", "config": {"workload": "OutlierAwareLinear.forward, fp16 activations, 16 outlier columns, bias", "M": M, "N": N, "K": K},
                "roofline": {"bound": "mfma", "achieved": round(tops, 1), "peak": PEAK_TFLOPS["int8"], "unit": "TOP/s",
                             "frac": round(tops / PEAK_TFLOPS["int8"], 4), "traffic": _outlier_traffic(), "kernel_us": round(kern_ms * 1e3, 2),
                             "note": "whole forward (2 kernels: masked row-wise quantiser + int8 GEMM with the outlier term and bias in its epilogue) / int8 dense peak; traffic: PMC bytes of the GEMM launch (profiles/traffic.json)"}})
    if not args.no_cpu_baseline:
        rows = 256
        xs = x[:rows].cpu()
        q, sc, oi, ow, b = oa.weight_int8.cpu(), oa.weight_scales.cpu(), oa.outlier_indices.cpu(), oa.outlier_weights.cpu(), oa.bias.cpu()
        t0 = time.perf_counter()
        oracle.outlier_linear(xs, q, sc, oi, ow, b)
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(2.0 * rows * N * K / dtc / 1e12, 4), "unit": "TOP/s", "cores": oracle.num_threads(),
                               "kind": "port", "sample": f"{rows} of the {M} rows, one pass (oracle, OpenMP)"}
    return out


def cpu_baseline(args, M, N, K, blocksize, compress, dtype):
    """The oracle's matmul_4bit (dequantize -> f32-accumulate GEMM: the reference's CPU algorithm,
    functional.py:752-773) on this box's host cores.  Sample: the same [M, K] x [N, K]^T workload
    (same synthetic weight), repeated until ~cpu_seconds of CPU time have been spent."""
    import oracle
    from mps_bitsandbytes_amd import synthetic
    threads = oracle.num_threads()
    W = synthetic.normal((N, K), dtype, seed=1234)
    packed, absmax, st2 = oracle.quantize_4bit(W, blocksize, "nf4", compress)
    X = synthetic.normal((M, K), dtype, seed=4321)
    oracle.matmul_4bit(X[:64], packed, absmax, (N, K), blocksize, "nf4", dtype, None, None, st2)  # warm the pool
    reps, spent = 0, 0.0
    while spent < args.cpu_seconds and reps < 64:
        t0 = time.perf_counter()
        oracle.matmul_4bit(X, packed, absmax, (N, K), blocksize, "nf4", dtype, None, None, st2)
        spent += time.perf_counter() - t0
        reps += 1
    return {"value": round(2.0 * M * N * K * reps / spent / 1e12, 5), "unit": "TFLOP/s", "cores": threads, "kind": "port",
            "sample": f"{reps} x the full step (M={M} rows, {N}x{K} weight dequantized per call), {spent:.1f} s of CPU time, "
                      f"OpenMP {threads} threads on {os.cpu_count()} logical cpus; C port of the reference's CPU path (oracle/)"}




def cpu_baseline_gemv(args, N, K, dtype):
    """M = 1 decode shape on the host cores: the oracle's matmul_4bit on one activation row, rotating over 8 layers
    (75 MB of packed weights + absmax, beyond the host's L2) for ~cpu_seconds; GB/s of the same algorithmic bytes."""
    import oracle
    from mps_bitsandbytes_amd import synthetic
    threads = oracle.num_threads()
    layers = []
    for i in range(8):
        W = synthetic.normal((N, K), dtype, seed=1234 + i)
        layers.append(oracle.quantize_4bit(W, 64, "nf4", False))
    X = synthetic.normal((1, K), dtype, seed=4321)
    nbytes = N * K // 2 + N * (K // 64) * 4 + K * 2 + N * 2
    calls, spent = 0, 0.0
    while spent < min(args.cpu_seconds, 10.0) and calls < 4096:
        t0 = time.perf_counter()
        for packed, absmax, st2 in layers:
            oracle.matmul_4bit(X, packed, absmax, (N, K), 64, "nf4", dtype, None, None, st2)
        spent += time.perf_counter() - t0
        calls += len(layers)
    return {"value": round(nbytes * calls / spent / 1e9, 3), "unit": "GB/s", "cores": threads, "kind": "port",
            "sample": f"{calls} M=1 calls over 8 rotating {N}x{K} layers, {spent:.1f} s of CPU time, OpenMP {threads} threads; "
                      f"C port of the reference's CPU path (oracle/)"}


def cpu_baseline_int8(args, M, N, K):
    """matmul_int8 on the host cores: the oracle's restatement of functional.py:788-793 on a row sample of the workload."""
    import oracle
    threads = oracle.num_threads()
    from mps_bitsandbytes_amd import synthetic
    rows = 256
    A = synthetic.int8_tensor((rows, K), seed=4321)
    B = synthetic.int8_tensor((K, N), seed=1234)
    sa = synthetic.normal((rows,), torch.float32, seed=77).abs() + 0.5
    sb = synthetic.normal((N,), torch.float32, seed=78).abs() + 0.5
    oracle.matmul_int8(A[:16], B, sa[:16], sb, torch.float16)
    reps, spent = 0, 0.0
    while spent < min(args.cpu_seconds, 10.0) and reps < 256:
        t0 = time.perf_counter()
        oracle.matmul_int8(A, B, sa, sb, torch.float16)
        spent += time.perf_counter() - t0
        reps += 1
    return {"value": round(2.0 * rows * N * K * reps / spent / 1e12, 5), "unit": "TOP/s", "cores": threads, "kind": "port",
            "sample": f"{reps} x {rows} of the {M} rows against the full {K}x{N} B, {spent:.1f} s of CPU time, OpenMP {threads} threads (oracle/)"}


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (torch.distributed.run, one
    per GPU) before this process has made any GPU call, relay their output and exit with their status."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    # stdout (rank 0's JSON line) passes straight through; stderr is relayed AND kept, so that a run that dies (a first 8-GPU run
    # nobody could rehearse) leaves its reason in the one line the driver records
    proc = subprocess.run(cmd, env=env, stderr=subprocess.PIPE, text=True, errors="replace")
    if proc.stderr:
        sys.stderr.write(proc.stderr)
        sys.stderr.flush()
    if proc.returncode != 0:
        print(json.dumps({"error": f"the {args.gpus} child ranks exited with status {proc.returncode}", "n_gpus": args.gpus,
                          "cmd": " ".join(cmd[1:8]) + " ...", "stderr_tail": (proc.stderr or "")[-1500:]}), flush=True)
    sys.exit(proc.returncode)


class Timer:
    """The timed region of the contract: barrier + synchronize, K steps, [finish], synchronize + barrier, MAX over
    ranks -- repeated `reps` times; HIP events recorded on the launching stream inside each repetition give the
    device-side duration of the same K launches."""

    def __init__(self, distributed, dist, reduce_dev):
        self.distributed, self.dist, self.reduce_dev = distributed, dist, reduce_dev

    def _fence(self):
        torch.cuda.synchronize()
        if self.distributed:
            self.dist.barrier()
        torch.cuda.synchronize()

    def prewarm(self, step, finish, ms):
        t0 = time.perf_counter()
        n = 0
        while (time.perf_counter() - t0) * 1e3 < ms:
            for _ in range(8):
                step()
            n += 8
            if finish is not None:
                finish()
            torch.cuda.synchronize()
        return n

    def run(self, step, finish, steps, reps):
        out = []
        st = torch.cuda.current_stream()
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self._fence()
            t0 = time.perf_counter()
            e0.record(st)
            for _ in range(steps):
                step()
            e1.record(st)
            if finish is not None:
                finish()
            self._fence()
            elapsed = time.perf_counter() - t0
            ev_ms = e0.elapsed_time(e1)
            if self.distributed:
                t = torch.tensor([elapsed, ev_ms], dtype=torch.float64, device=self.reduce_dev)
                self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
                elapsed, ev_ms = float(t[0].item()), float(t[1].item())
            out.append((elapsed, ev_ms))
        return out


def median_rep(reps):
    """(wall seconds, event ms) of the repetition with the median wall time"""
    order = sorted(range(len(reps)), key=lambda i: reps[i][0])
    return reps[order[len(order) // 2]]


class Ctx:
    """what measure() needs of the process: rank / world, device, torch.distributed (or None), the timer"""
    pass


def gen_normal(synthetic, shape, dt, seed, std, dev):
    """bench inputs from the repo's counter-based PRNG, generated on the device (synthetic.normal_device: bit-identical
    to synthetic.normal on the host)"""
    return synthetic.normal_device(shape, dt, seed=seed, std=std, device=dev)


def measure(args, wl, ctx):
    """One workload -> the dict of its bench line (rank 0; other ranks take part in the timed region and return their copy)."""
    import mps_bitsandbytes_amd as bnb
    from mps_bitsandbytes_amd import _native, synthetic
    from mps_bitsandbytes_amd.sharding import ChunkedGather, row_shard
    world, rank, dev, dist, distributed, rehearse = ctx.world, ctx.rank, ctx.dev, ctx.dist, ctx.distributed, ctx.rehearse
    if wl == "nf4_m4096":
        M, N, K, dt, compress, name = 4096, 4096, 4096, torch.bfloat16, False, "bf16"
    elif wl == "nf4dq_ffn":
        M, N, K, dt, compress, name = 4096, 11008, 4096, torch.bfloat16, True, "bf16"
    elif wl == "nf4_m1":
        M, N, K, dt, compress, name = 1, 4096, 4096, torch.float16, False, "f16"
    else:
        M, N, K, dt, compress, name = 4096, 4096, 4096, torch.float16, False, "int8"

    M_global = M * world
    s_row, e_row = row_shard(M_global, rank, world)
    out = {"metric": None, "value": None, "unit": None, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": name,
           "data": "synthetic (repo PRNG: synthetic.py, weights seed 1234 + layer, activations seed 4321 + rank)"}
    timer = Timer(distributed, dist, "cpu" if rehearse else dev)
    reps_n = max(1, args.reps)

    do_gather = distributed and not args.no_gather and wl in ("nf4_m4096", "nf4dq_ffn")
    modes = {}        # name -> (step, finish)
    if wl in ("nf4_m4096", "nf4dq_ffn", "nf4_m1"):
        std = 1.0 if wl != "nf4dq_ffn" else 0.02
        if wl == "nf4_m1":
            # rotate over 64 distinct layers so the weights come from HBM, not the Infinity Cache
            layers = []
            for i in range(64):
                Wi = gen_normal(synthetic, (N, K), dt, 1234 + i, 1.0, dev)
                layers.append(bnb.quantize_nf4(Wi, blocksize=64))
                del Wi
            X = gen_normal(synthetic, (1, K), dt, 4321 + rank, 1.0, dev)

            def eager_pass():
                for p, st in layers:
                    bnb.matmul_4bit(X, p, st)
            # one HIP graph of the 64 launches: the per-layer kernel (~5 us) is shorter than a Python-issued launch, so an
            # eager loop would time the host
            for _ in range(2):
                eager_pass()
            torch.cuda.synchronize()
            m1_graph = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                eager_pass()
                with torch.cuda.graph(m1_graph, stream=side):
                    eager_pass()
            torch.cuda.current_stream().wait_stream(side)
            modes["main"] = (m1_graph.replay, None)
            flops_per_step = 2.0 * N * K * 64
            bytes_per_launch = N * K // 2 + N * (K // 64) * 4 + K * 2 + N * 2   # 9 453 568 B
        else:
            W = gen_normal(synthetic, (N, K), dt, 1234, std, dev)     # the replicated weight: the same on every rank
            packed, state = bnb.quantize_nf4(W, blocksize=64, compress_statistics=compress)
            del W
            X = gen_normal(synthetic, (e_row - s_row, K), dt, 4321 + rank, 1.0, dev)   # this rank's rows of the global batch

            def gemm_step():
                bnb.matmul_4bit(X, packed, state)
            flops_per_step = 2.0 * M * N * K
            if not do_gather:
                modes["main"] = (gemm_step, None)
            else:
                # The three curves of SURVEY 8e (+ the row-chunked form), all in one line:
                #   gemm_only   no exchange: communication-free scaling of the sharded GEMM
                #   sync        blocking all-gather after every GEMM
                #   overlapped  async all-gather of step i (RCCL's stream) under the GEMM of step i+1 (two result buffers)
                #   chunked     inside ONE step: the rank's rows in c chunks, GEMM of chunk i+1 under the gather of chunk i
                gathered = [torch.empty(M_global, N, dtype=dt, device=dev) for _ in range(2)]
                pending = [None, None]     # (work handle, the shard it reads): the shard stays referenced until its gather is waited for
                counter = [0]

                def overlapped_step():
                    y = bnb.matmul_4bit(X, packed, state)
                    i = counter[0] & 1
                    counter[0] += 1
                    if pending[i] is not None:
                        pending[i][0].wait()
                    pending[i] = (dist.all_gather_into_tensor(gathered[i], y, async_op=True), y)

                def overlapped_finish():
                    for i in range(2):
                        if pending[i] is not None:
                            pending[i][0].wait()
                            pending[i] = None

                def sync_step():
                    y = bnb.matmul_4bit(X, packed, state)
                    dist.all_gather_into_tensor(gathered[0], y)

                cg = ChunkedGather(lambda x: bnb.matmul_4bit(x, packed, state), e_row - s_row, N, dt, dev, world,
                                   chunks=args.chunks)
                modes["gemm_only"] = (gemm_step, None)
                modes["sync"] = (sync_step, None)
                modes["overlapped"] = (overlapped_step, overlapped_finish)
                modes["chunked"] = (lambda: cg.step(X), cg.finish)
                modes["main"] = modes["sync"] if args.sync_gather else modes["overlapped"]
    else:
        # A, B ~ N(0, 1) fp16, quantised by quantize_rowwise (A by row, B by column)
        A16 = gen_normal(synthetic, (M, K), dt, 4321 + rank, 1.0, dev)
        B16 = gen_normal(synthetic, (K, N), dt, 1234, 1.0, dev)
        A, sa = bnb.quantize_rowwise(A16)
        Bt, sb = bnb.quantize_rowwise(B16.t().contiguous())
        B = Bt.t().contiguous()          # [K, N] as the reference passes it
        del A16, B16, Bt

        def step():
            bnb.matmul_int8(A, B, sa, sb, torch.float16)
        modes["main"] = (step, None)
        flops_per_step = 2.0 * M * N * K

    step, finish = modes["main"]
    # the driver's protocol WITHOUT the time-based pre-warm, measured first (cold clock): W warm-up steps, then K timed steps
    for _ in range(args.warmup):
        step()
    if finish is not None:
        finish()
    cold = timer.run(step, finish, args.steps, 1)[0]
    prewarm_steps = timer.prewarm(step, finish, args.prewarm_ms) if args.prewarm_ms > 0 else 0
    for _ in range(args.warmup):
        step()
    if finish is not None:
        finish()
    reps = timer.run(step, finish, args.steps, reps_n)
    elapsed, ev_ms = median_rep(reps)
    kernel_name = _native.last_kernel()
    out["prewarm_ms"] = args.prewarm_ms
    out["prewarm_steps"] = prewarm_steps
    out["repetitions"] = reps_n
    out["rep_ms_per_step"] = [round(r[0] / args.steps * 1e3, 5) for r in reps]

    curves = {}
    if do_gather:
        for cname in ("gemm_only", "sync", "overlapped", "chunked"):
            if modes[cname] is modes["main"]:
                curves[cname] = (elapsed, ev_ms)
                continue
            s_, f_ = modes[cname]
            for _ in range(max(2, args.warmup)):
                s_()
            if f_ is not None:
                f_()
            curves[cname] = median_rep(timer.run(s_, f_, args.steps, min(reps_n, 3)))

    # device-side duration of one step: the events of the median repetition (for N > 1: of the gather-free curve)
    if do_gather:
        kern_ms = curves["gemm_only"][1] / args.steps
    elif wl == "nf4_m1":
        kern_ms = ev_ms / args.steps / 64
    else:
        kern_ms = ev_ms / args.steps

    ms_per_step = elapsed / args.steps * 1e3
    total_flops = flops_per_step * world
    out["ms_per_step"] = round(ms_per_step, 5)
    par = f"rows sharded x{world}, weights replicated"
    if do_gather:
        par += (", all-gather of outputs (RCCL, blocking)" if args.sync_gather
                else ", all-gather of outputs (RCCL, async: the gather of step i overlaps the GEMM of step i+1)")
    two_launch = kernel_name.startswith("dequant+dense")
    how = ("dequantize_4bit ONCE into a transient scratch + dense MFMA GEMM (two launches per step; the reference's own two steps above M = 512)"
           if two_launch else "one fused dequant + MFMA launch per step")
    out["config"] = {"workload": {"nf4_m4096": f"Linear4bit-style NF4 matmul_4bit, weight 4096x4096 bf16-origin bs64, M=4096 rows per GPU: {how}",
                                  "nf4dq_ffn": f"NF4 + double-quant absmax matmul_4bit, weight 11008x4096 bf16 bs64, M=4096: {how}",
                                  "int8_4096": "rowwise INT8 matmul_int8 4096x4096x4096 on int8 MFMA (A by row, B [K, N] by column, read in place)",
                                  "nf4_m1": "fused NF4 dequant+GEMV, weight 4096x4096 fp16 bs64, M=1, rotating over 64 layers (one HIP graph of the 64 launches per step)"}[wl],
                     "global_rows": M_global, "rows_per_gpu": M, "N": N, "K": K, "parallelism": par, "kernel": kernel_name}
    out["no_prewarm"] = {"ms_per_step": round(cold[0] / args.steps * 1e3, 5),
                         "note": "the same K steps after the W warm-up steps only, BEFORE the time-based pre-warm (cold clock): what a driver-side "
                                 "timer around a fresh process sees without the disclosed pre-warm"}
    traffic = {}
    prof = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(prof):
        try:
            traffic = json.load(open(prof))
        except Exception:
            traffic = {}
    # `traffic` figures are PMC counters collected by tools/profile_round.sh (rocprofv3 --pmc cannot run inside this process);
    # the file says which build and day they are from, and the line repeats it so that a stale file is visible
    src_ = traffic.get("_source", {}) if isinstance(traffic.get("_source"), dict) else {}
    traffic_source = f"profiles/traffic.json: commit {src_.get('commit', 'unrecorded')}, {src_.get('date', 'undated')}, {src_.get('tag', '')}".strip(", ")
    if wl == "nf4_m1":
        gbs = bytes_per_launch / (kern_ms * 1e-3) / 1e9
        out["metric"] = "effective GB/s, fused NF4 dequant+GEMV 4096x4096 M=1 (HBM, 64 rotating layers)"
        out["value"] = round(bytes_per_launch * 64 * world / (elapsed / args.steps) / 1e9, 2)
        out["unit"] = "GB/s"
        out["no_prewarm"]["value"] = round(bytes_per_launch * 64 * world / (cold[0] / args.steps) / 1e9, 2)
        out["roofline"] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                           "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": traffic.get("k_gemv4_bytes_per_launch"),
                           "kernel_us": round(kern_ms * 1e3, 3), "traffic_source": traffic_source}
    else:
        tflops = total_flops / (elapsed / args.steps) / 1e12
        peak = PEAK_TFLOPS[name]
        kern_tflops = flops_per_step / (kern_ms * 1e-3) / 1e12
        unit = "TOP/s" if wl == "int8_4096" else "TFLOP/s"
        out["metric"] = {"nf4_m4096": "effective bf16 TFLOPS, NF4 dequant+matmul 4096x4096 @ M=4096 (" + ("dequantise-once + dense GEMM, two launches" if two_launch else "one fused launch") + ")",
                         "nf4dq_ffn": "effective bf16 TFLOPS, NF4 + double-quant matmul 11008x4096 @ M=4096 (" + ("dequantise-once + dense GEMM, two launches" if two_launch else "one fused launch") + ")",
                         "int8_4096": "effective int8 TOPS, rowwise matmul_int8 4096x4096x4096"}[wl]
        out["value"] = round(tflops, 2)
        out["unit"] = unit
        out["no_prewarm"]["value"] = round(total_flops / (cold[0] / args.steps) / 1e12, 2)
        for cname, (el, _) in curves.items():
            out[cname] = {"value": round(total_flops / (el / args.steps) / 1e12, 2), "unit": unit,
                          "ms_per_step": round(el / args.steps * 1e3, 5)}
        if do_gather:
            out["curves_note"] = ("gemm_only = no exchange; sync = blocking all-gather per step; overlapped = async gather of step i "
                                  f"under the GEMM of step i+1; chunked = {args.chunks} row chunks per step, GEMM of chunk i+1 under "
                                  "the gather of chunk i (each chunk fills only 1/c of the 256 CUs at this shape)")
        # The roofline object describes the OPERATION (the whole step: every launch of it), per GPU: algorithmic flops / step
        # duration by HIP events on the launching stream.  `traffic`: PMC bytes of the step's launches (profiles/traffic.json).
        tkey = {"nf4_m4096": "nf4_m4096_step_bytes", "nf4dq_ffn": "nf4dq_ffn_step_bytes", "int8_4096": "int8_4096_step_bytes"}[wl]
        out["roofline"] = {"bound": "mfma", "scope": "the whole step (all launches of one matmul call)", "achieved": round(kern_tflops, 2),
                           "peak": peak, "unit": unit, "frac": round(kern_tflops / peak, 4), "traffic": traffic.get(tkey),
                           "kernel_us": round(kern_ms * 1e3, 2), "traffic_source": traffic_source,
                           "kernel_us_note": "HIP events around the K steps of the median repetition / K (includes the inter-launch boundaries)"}
        lib, sp = _native.lib(), _native.stream_ptr(dev)
        st_ = torch.cuda.current_stream()

        def ev_us(fn):
            vals = []
            for _ in range(reps_n):
                for _ in range(3):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st_)
                for _ in range(args.steps):
                    fn()
                e1.record(st_)
                e1.synchronize()
                vals.append(e0.elapsed_time(e1) / args.steps * 1e3)
            return sorted(vals)[len(vals) // 2]

        if two_launch and wl in ("nf4_m4096", "nf4dq_ffn"):
            # the step's two launches on their own (same operands, same stream, events around K launches)
            code = _native.DTYPE_CODE[dt]
            Wd = bnb.dequantize_4bit(packed, state)
            Yd = torch.empty(M, N, dtype=dt, device=dev)

            def dense_only():
                rc = lib.mbnb_gemm_dense(X.data_ptr(), Wd.data_ptr(), code, None, code, Yd.data_ptr(), M, N, K, K, None, 0, 1, sp)
                if rc != 0:
                    raise RuntimeError(f"mbnb_gemm_dense: {rc}")

            def dequant_only():
                bnb.dequantize_4bit(packed, state, out=Wd)

            dense_us, deq_us = ev_us(dense_only), ev_us(dequant_only)
            assert torch.equal(Yd, bnb.matmul_4bit(X, packed, state)), "dense-only launch differs from the step's output"
            d_tflops = flops_per_step / (dense_us * 1e-6) / 1e12
            out["roofline"]["launches"] = ["k_dequantize_4bit_flat (N x K_weight 16-bit values into the scratch)", "k_gemm_dense"]
            out["roofline"]["dominant_kernel"] = {
                "kernel": "k_gemm_dense", "kernel_us": round(dense_us, 2), "achieved": round(d_tflops, 2), "frac": round(d_tflops / peak, 4),
                "traffic": traffic.get("k_gemm_dense_bytes_per_launch") if wl == "nf4_m4096" else traffic.get("nf4dq_ffn_gemm_dense_bytes_per_launch"),
                "note": "k_gemm_dense ALONE on the already dequantised weight (HIP events around K launches / K): NOT the operation the metric names"}
            # the same three numbers as scalars of `roofline` (records that keep only an object's scalar members keep these)
            out["roofline"]["dom_kernel"] = "k_gemm_dense"
            out["roofline"]["dom_kernel_us"] = round(dense_us, 2)
            out["roofline"]["dom_frac"] = round(d_tflops / peak, 4)
            out["roofline"]["dequantize_us"] = round(deq_us, 2)
            out["roofline"]["dequantize_us_note"] = ("the public dequantize_4bit launch ALONE (eager call: includes its host side); inside the step the pass stores write-through "
                                                     "(four dwords per thread on weights of up to 32 Mi elements), see step_minus_dense_us")
            out["roofline"]["step_minus_dense_us"] = round(out["roofline"]["kernel_us"] - dense_us, 2)
            del Wd, Yd
        if wl == "int8_4096":
            if kernel_name == "i8_inplace4":     # one launch: the operation IS its dominant kernel
                out["roofline"]["dom_kernel"], out["roofline"]["dom_kernel_us"], out["roofline"]["dom_frac"] = "k_gemm_i8_inplace", out["roofline"]["kernel_us"], out["roofline"]["frac"]
            out["roofline"]["launches"] = (["k_gemm_i8_inplace (four waves, B [K, N] read in place: no transpose pass, no workspace)"] if kernel_name == "i8_inplace4"
                                           else ["k_transpose_i8 (B [K, N] -> [N, K] into the workspace)", "k_gemm_dense<I8>"] if "dense" in kernel_name else [kernel_name])

    if args.verify and do_gather:
        # gathered == unsharded: rank 0 rebuilds every rank's rows and runs the whole batch through the same kernel
        overlapped_finish()
        y_local = bnb.matmul_4bit(X, packed, state)
        full = torch.empty(M_global, N, dtype=dt, device=dev)
        dist.all_gather_into_tensor(full, y_local)
        chunked_full = cg.step(X)
        cg.finish()
        ok = None
        if rank == 0:
            xs = []
            for r in range(world):
                sr, er = row_shard(M_global, r, world)
                xs.append(gen_normal(synthetic, (er - sr, K), dt, 4321 + r, 1.0, dev))
            ref = bnb.matmul_4bit(torch.cat(xs), packed, state)
            ok = bool(torch.equal(full, ref))
            ch = chunked_full.reshape(M_global, N)
            if M // max(1, args.chunks) > 512:      # an unsplit dense product: a row's bits do not depend on the rows computed with it
                ok = ok and bool(torch.equal(ch, ref))
            else:                                    # <= 512 rows per chunk may take k_gemm_small (another summation order)
                ok = ok and float((ch.double() - ref.double()).norm() / ref.double().norm()) <= 1e-3
            del xs, ref
        out["verified"] = ok
        del full

    if rank == 0 and wl == "nf4_m4096":
        if not args.no_gemv:
            # the M = 1 half of the metric: HBM-bound decode shape, 64 rotating layers
            layers = []
            for i in range(64):
                Wi = gen_normal(synthetic, (N, K), dt, 2000 + i, 1.0, dev)
                layers.append(bnb.quantize_nf4(Wi, blocksize=64))
                del Wi
            x1 = gen_normal(synthetic, (1, K), dt, 4320, 1.0, dev)

            def gemv_pass():
                for p, st in layers:
                    bnb.matmul_4bit(x1, p, st)
            for _ in range(3):
                gemv_pass()
            torch.cuda.synchronize()
            # one HIP graph of the 64 back-to-back launches: the per-layer kernel (~2-3 us) is shorter
            # than a Python-issued launch, so eager timing would measure the host, not the GPU
            graph = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                gemv_pass()
                with torch.cuda.graph(graph, stream=side):
                    gemv_pass()
            torch.cuda.current_stream().wait_stream(side)
            for _ in range(10):
                graph.replay()
            torch.cuda.synchronize()
            us = sorted(event_time_ms(graph.replay, 20) for _ in range(5))[2] / 64 * 1e3
            # the single-hot-layer figure: the same 64 launches on ONE layer, whose 9.4 MB stay in the
            # 256 MiB Infinity Cache -- labelled as such, never the headline
            hot = torch.cuda.CUDAGraph()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                p0, st0 = layers[0]
                bnb.matmul_4bit(x1, p0, st0)
                with torch.cuda.graph(hot, stream=side):
                    for _ in range(64):
                        bnb.matmul_4bit(x1, p0, st0)
            torch.cuda.current_stream().wait_stream(side)
            for _ in range(5):
                hot.replay()
            torch.cuda.synchronize()
            us_hot = sorted(event_time_ms(hot.replay, 20) for _ in range(5))[2] / 64 * 1e3
            nbytes = N * K // 2 + N * (K // 64) * 4 + K * 2 + N * 2
            gbs = nbytes / (us * 1e-6) / 1e9
            out["gemv"] = {"workload": "fused NF4 dequant+GEMV 4096x4096 M=1 bf16, 64 rotating layers (605 MB), one HIP graph of 64 launches (per-layer time includes the ~1 us launch boundary); median of 5 x 20 replays",
                           "kernel": _native.last_kernel(), "us_per_layer": round(us, 3), "bytes_per_layer": nbytes,
                           "us_per_layer_hot": round(us_hot, 3),
                           "hot_note": "same launches on ONE layer (weights resident in the Infinity Cache): not the headline",
                           "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                        "frac": round(gbs / PEAK_HBM_GBS, 4),
                                        "traffic": traffic.get("k_gemv4_bytes_per_launch")}}
            del layers
        if not args.no_gemv:
            # batch sizes between the two metric shapes (the reference's native path serves M <= 512): one HIP graph of
            # 24 calls per M (the graph's own launch, ~10 us per replay, is then < 0.5 us of a call), same weight; which kernel served it is recorded
            # next to the time
            sweep = []
            for Ms in (2, 16, 64, 128, 256, 512, 1024):
                xs_ = gen_normal(synthetic, (Ms, K), dt, 5000 + Ms, 1.0, dev)
                gr = torch.cuda.CUDAGraph()
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    bnb.matmul_4bit(xs_, packed, state)
                    with torch.cuda.graph(gr, stream=side):
                        for _ in range(24):
                            bnb.matmul_4bit(xs_, packed, state)
                torch.cuda.current_stream().wait_stream(side)
                for _ in range(3):
                    gr.replay()
                torch.cuda.synchronize()
                us = event_time_ms(gr.replay, 10) / 24 * 1e3
                sweep.append({"M": Ms, "us": round(us, 2), "kernel": _native.last_kernel()})
            out["batch_sweep"] = sweep
        if not args.no_empirical:
            # Empirical ceilings of this box next to the vendor peaks: the vendor BLAS on the same
            # 4096^3 bf16 problem with the weight already dequantised (torch.matmul -> hipBLASLt/rocBLAS; context
            # only, not part of the product) and a 1 GiB device-to-device copy.
            try:
                Wd = bnb.dequantize_4bit(packed, state)
                torch.matmul(X, Wd.t())
                blas_ms = min(event_time_ms(lambda: torch.matmul(X, Wd.t()), 50) for _ in range(3))
                del Wd
                src = torch.empty(1 << 28, dtype=torch.float32, device=dev)
                dst = torch.empty_like(src)
                dst.copy_(src)
                copy_ms = min(event_time_ms(lambda: dst.copy_(src), 10) for _ in range(3))
                del src, dst
                # what the matrix pipe sustains on THIS box: a bare bf16 MFMA loop on every SIMD (tools/libmbnb_probe.so); the
                # spec peak assumes 2.4 GHz, the chip holds much less under MFMA load
                mfma_tflops = bare_mfma_rate(0)
                out["roofline"]["empirical"] = {
                    "vendor_blas_bf16_same_shape_tflops": round(2.0 * M * N * K / (blas_ms * 1e-3) / 1e12, 1),
                    "vendor_blas_us": round(blas_ms * 1e3, 1),
                    "bare_mfma_loop_bf16_tflops": None if mfma_tflops is None else round(mfma_tflops, 1),
                    "bare_mfma_note": "v_mfma_f32_32x32x16_bf16 back to back on every SIMD, nothing else: the sustained matrix rate at the clock the chip holds under MFMA load",
                    "dtod_copy_gbs_read_plus_write": round(2.0 * (1 << 30) / (copy_ms * 1e-3) / 1e9, 0)}
                # The distance to the 0.60 target against MEASURED limits of this box: the bare matrix rate, and the
                # fill rate of the two 33.5 MB writes the operation cannot avoid (the scratch by the dequantise pass, the output by the
                # epilogue -- one tile per CU: every workgroup stores at the end, nothing left to overlap them with).
                fillbuf = torch.empty(M * N, dtype=dt, device=dev)
                fillbuf.zero_()
                fill_us = min(event_time_ms(fillbuf.zero_, 50) for _ in range(3)) * 1e3
                del fillbuf
                if mfma_tflops:
                    mfma_us = 2.0 * M * N * K / (mfma_tflops * 1e12) * 1e6
                    floor_us = mfma_us + 2.0 * fill_us
                    out["roofline"]["ceiling"] = {
                        "bare_mfma_tflops": round(mfma_tflops, 1), "bare_mfma_frac_of_peak": round(mfma_tflops / peak, 4),
                        "mfma_only_us": round(mfma_us, 2), "fill_33MB_us": round(fill_us, 2),
                        "floor_us": round(floor_us, 2), "floor_frac": round(2.0 * M * N * K / (floor_us * 1e-6) / 1e12 / peak, 4),
                        "kstep_cycles_per_2048_mfma_cycles": 2268, "clock_ghz_in_kernel": 1.76,
                        "note": "floor = every MFMA of the product at this box's bare-loop rate + the scratch write + the output write at this box's fill rate "
                                "(torch zero_ of 33.5 MB, back to back), nothing else; k-step cycles and in-kernel clock: s_memtime stamps, "
                                "profiles/r03_dense_kstep_stamps.txt (the k-loop is 0.91 MFMA-busy and waits for nothing)"}
                    out["roofline"]["ceiling_floor_us"] = round(floor_us, 2)
                    out["roofline"]["ceiling_floor_frac"] = out["roofline"]["ceiling"]["floor_frac"]
            except Exception as e:  # context only: never fails the bench
                out["roofline"]["empirical"] = {"error": str(e)[:200]}
    if rank == 0 and wl == "int8_4096" and not args.no_empirical:
        # what the int8 matrix pipe sustains on THIS box (bare v_mfma_i32_32x32x32_i8 loop, one wave per SIMD)
        try:
            i8_tops = bare_mfma_rate(1)
            if i8_tops is not None:
                out["roofline"]["empirical"] = {
                    "bare_mfma_loop_i8_tops": round(i8_tops, 1),
                    "bare_mfma_note": "v_mfma_i32_32x32x32_i8 back to back on every SIMD, nothing else: the sustained int8 matrix rate at the clock the chip holds under that load"}
        except Exception as e:  # context only
            out["roofline"]["empirical"] = {"error": str(e)[:200]}
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        import oracle
        nproc = oracle.num_threads()
        base = {"nf4_m4096": lambda: cpu_baseline(args, M, N, K, 64, compress, dt), "nf4dq_ffn": lambda: cpu_baseline(args, M, N, K, 64, compress, dt),
                "nf4_m1": lambda: cpu_baseline_gemv(args, N, K, dt), "int8_4096": lambda: cpu_baseline_int8(args, M, N, K)}[wl]
        out["cpu_baseline"] = base()
        if nproc > 8:     # the host baseline at nproc threads AND at 8 threads (the build container's core count)
            oracle.set_num_threads(8)
            try:
                half = argparse.Namespace(**vars(args))
                half.cpu_seconds = max(2.0, args.cpu_seconds / 2)
                b8 = {"nf4_m4096": lambda: cpu_baseline(half, M, N, K, 64, compress, dt), "nf4dq_ffn": lambda: cpu_baseline(half, M, N, K, 64, compress, dt),
                      "nf4_m1": lambda: cpu_baseline_gemv(half, N, K, dt), "int8_4096": lambda: cpu_baseline_int8(half, M, N, K)}[wl]()
                out["cpu_baseline_8_threads"] = b8
            finally:
                oracle.set_num_threads(nproc)
    return out


def digest(out):
    """All four GPU configs of BASELINE.json in <= 600 characters: [value, roofline frac of the operation, step us, dominant-kernel us]
    per workload (nf4_m1: [GB/s, frac, us per layer]); batch_sweep_us: M -> us of matmul_4bit at 4096 x 4096."""
    def row(o):
        r = o.get("roofline") or {}
        return [o.get("value"), r.get("frac"), r.get("kernel_us"), r.get("dom_kernel_us")]
    d = {"nf4_m4096": row(out)}
    for o in out.get("secondary", []):
        key = o.get("workload_key")
        if "error" in o:
            d[key] = "error"
        elif key == "nf4_m1":
            d[key] = [o.get("value"), (o.get("roofline") or {}).get("frac"), (o.get("roofline") or {}).get("kernel_us")]
        else:
            d[key] = row(o)
    if "gemv" in out:
        d["gemv_bf16"] = [out["gemv"]["roofline"]["achieved"], out["gemv"]["roofline"]["frac"], out["gemv"]["us_per_layer"]]
    if "batch_sweep" in out:
        d["batch_sweep_us"] = {str(e["M"]): e["us"] for e in out["batch_sweep"]}
    return d


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)   # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(json.dumps({"error": f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU"}))
        sys.exit(2)
    distributed = world > 1
    if distributed:
        # RCCL shares device buffers between the ranks through dmabuf IPC on this driver; the legacy mode fails with hipIpcGetMemHandle: invalid argument.
        # (Set before the first HIP call of the process; a launcher's own setting wins.)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        print(json.dumps({"error": "no GPU visible; bench.py measures the HIP path only", "rank": rank, "world": world}), flush=True)
        sys.exit(2)
    # BENCH_REHEARSE=1: every rank uses cuda:0 and the gloo backend -- lets the N > 1 code path be exercised on a
    # one-GPU box (RCCL refuses two ranks on one device); numbers from such a run mean nothing.
    rehearse = os.environ.get("BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import mps_bitsandbytes_amd as bnb
    from mps_bitsandbytes_amd import _native, synthetic
    _native.lib()  # fail loudly when the HIP library is missing

    wl = args.workload
    if wl in ("embed4", "embed8", "outlier", "w8a16", "fp8"):
        if rank == 0:
            print(json.dumps(bench_nn(args, wl, dev, bnb, synthetic)))
        return
    ctx = Ctx()
    ctx.world, ctx.rank, ctx.dev, ctx.dist, ctx.distributed, ctx.rehearse = world, rank, dev, dist, distributed, rehearse
    out = measure(args, wl, ctx)
    if rank == 0 and world == 1 and wl == "nf4_m4096" and not args.no_secondary:
        # the other BASELINE GPU configs under the same clock: configs[2] (11008 x 4096 + double quant), configs[3] (int8 4096^3),
        # configs[1] stand-alone (M = 1 over 64 rotating layers), each with its own roofline / traffic / cpu_baseline
        sec = []
        for w2 in ("nf4dq_ffn", "int8_4096", "nf4_m1"):
            a2 = argparse.Namespace(**vars(args))
            a2.cpu_seconds = min(args.cpu_seconds, 5.0)
            a2.prewarm_ms = min(args.prewarm_ms, 200.0)
            a2.reps = min(args.reps, 3)
            try:
                o2 = measure(a2, w2, ctx)
                o2["workload_key"] = w2
                sec.append(o2)
            except Exception as e:   # a secondary line never costs the headline
                sec.append({"workload_key": w2, "error": str(e)[:300]})
            torch.cuda.empty_cache()
        out["secondary"] = sec
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        out["digest"] = digest(out)      # LAST key: survives a record that keeps only the tail of the line
        print(json.dumps(out))


if __name__ == "__main__":
    main()
