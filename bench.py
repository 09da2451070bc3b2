#!/usr/bin/env python3
"""
bench.py — throughput of the fused NF4 dequant + matmul hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: Linear4bit-style fused NF4 dequant + matmul,
weight 4096x4096 (bf16-origin, blocksize 64), M = 4096 rows per GPU, bf16 activations, inputs
resident in HBM.  With N > 1 the weight is replicated, every rank owns 4096 rows of a global
batch of 4096*N (BASELINE configs[4] at N = 8: M = 32768) and the step ends with an RCCL
all-gather of the output shards over xGMI (weak scaling).

Rank 0 prints ONE JSON line: metric/value per the driver contract plus
  "roofline"     the dominant kernel (k_gemm_decode, MFMA-bound): algorithmic flops per launch /
                 average launch duration measured with HIP events on the launching stream
  "cpu_baseline" the CPU oracle (a port of the reference's CPU path) timed on this box's host
                 cores on a bounded row-sample of the same workload
  "gemv"         the M = 1 decode shape of the metric (HBM-bound), rotating over 64 distinct
                 layers (605 MB > 256 MB Infinity Cache), with its own roofline fraction.
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
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample")
    return ap.parse_args()


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
                    "config": {"workload": f"{name}.forward, bf16 activations, weights decoded inside the 256 x 256 LDS-DMA GEMM", "M": M, "N": N, "K": K,
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
                             "frac": round(tops / PEAK_TFLOPS["int8"], 4), "traffic": None, "kernel_us": round(kern_ms * 1e3, 2),
                             "note": "whole forward (3 kernels) / int8 dense peak"}})
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


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if not torch.cuda.is_available():
        print(json.dumps({"error": "no GPU visible; bench.py measures the HIP path only"}))
        sys.exit(2)
    # BENCH_REHEARSE=1: every rank uses cuda:0 and the gloo backend -- lets the N > 1 code path be exercised on a
    # one-GPU box (RCCL refuses two ranks on one device); numbers from such a run mean nothing.
    rehearse = os.environ.get("BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import mps_bitsandbytes_amd as bnb
    from mps_bitsandbytes_amd import _native, synthetic
    from mps_bitsandbytes_amd.sharding import row_shard
    _native.lib()  # fail loudly when the HIP library is missing

    wl = args.workload
    if wl in ("embed4", "embed8", "outlier", "w8a16", "fp8"):
        if rank == 0:
            print(json.dumps(bench_nn(args, wl, dev, bnb, synthetic)))
        return
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
    g = torch.Generator(device=dev)
    g.manual_seed(4321 + rank)

    out = {"metric": None, "value": None, "unit": None, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": name,
           "data": "synthetic"}

    gathered = None
    if wl in ("nf4_m4096", "nf4dq_ffn", "nf4_m1"):
        std = 1.0 if wl != "nf4dq_ffn" else 0.02
        W = (torch.randn(N, K, generator=g, device=dev, dtype=torch.float32) * std).to(dt)
        if wl == "nf4_m1":
            # rotate over 64 distinct layers so the weights come from HBM, not the Infinity Cache
            layers = []
            for i in range(64):
                Wi = (torch.randn(N, K, generator=g, device=dev, dtype=torch.float32)).to(dt)
                layers.append(bnb.quantize_nf4(Wi, blocksize=64))
                del Wi
            X = torch.randn(1, K, generator=g, device=dev, dtype=torch.float32).to(dt)
            ys = torch.empty(64, N, dtype=dt, device=dev)

            def step():
                for p, st in layers:
                    bnb.matmul_4bit(X, p, st)
            flops_per_step = 2.0 * N * K * 64
            bytes_per_launch = N * K // 2 + N * (K // 64) * 4 + K * 2 + N * 2   # 9 453 568 B
        else:
            packed, state = bnb.quantize_nf4(W, blocksize=64, compress_statistics=compress)
            del W
            X = torch.randn(e_row - s_row, K, generator=g, device=dev, dtype=torch.float32).to(dt)
            if distributed and not args.no_gather:
                # two result buffers: the all-gather of step i (RCCL's own stream, async_op) overlaps the GEMM of
                # step i+1; a buffer is reused only after the gather that filled it two steps earlier has completed
                gathered = [torch.empty(M_global, N, dtype=dt, device=dev) for _ in range(2)]
            pending = [None, None]
            counter = [0]

            def step():
                y = bnb.matmul_4bit(X, packed, state)
                if gathered is not None:
                    i = counter[0] & 1
                    counter[0] += 1
                    if pending[i] is not None:
                        pending[i].wait()
                    if args.sync_gather:
                        dist.all_gather_into_tensor(gathered[i], y)
                    else:
                        pending[i] = dist.all_gather_into_tensor(gathered[i], y, async_op=True)
            flops_per_step = 2.0 * M * N * K
    else:
        A = torch.randint(-127, 128, (M, K), generator=g, device=dev, dtype=torch.int8)
        B = torch.randint(-127, 128, (K, N), generator=g, device=dev, dtype=torch.int8)
        sa = torch.rand(M, generator=g, device=dev) + 0.5
        sb = torch.rand(N, generator=g, device=dev) + 0.5

        def step():
            bnb.matmul_int8(A, B, sa, sb, torch.float16)
        flops_per_step = 2.0 * M * N * K

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if gathered is not None:
        for w in pending:
            if w is not None:
                w.wait()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_name = _native.last_kernel()

    # N > 1 with the gather: a second timed region WITHOUT it (same bracket: barrier + synchronize, max over ranks), so
    # one run carries both the headline (GEMM + all-gather) and the communication-free scaling of the sharded GEMM
    gemm_only = None
    if gathered is not None:
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            bnb.matmul_4bit(X, packed, state)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        e2 = time.perf_counter() - t1
        t = torch.tensor([e2], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        gemm_only = float(t.item())

    # kernel-only launch duration (no gather, no host gaps beyond back-to-back launches)
    if wl == "nf4_m1":
        kern_ms = event_time_ms(step, max(1, args.steps // 4)) / 64
    elif wl == "int8_4096":
        kern_ms = event_time_ms(step, args.steps)
    else:
        kern_ms = event_time_ms(lambda: bnb.matmul_4bit(X, packed, state), args.steps)

    ms_per_step = elapsed / args.steps * 1e3
    total_flops = flops_per_step * world
    out["ms_per_step"] = round(ms_per_step, 5)
    out["config"] = {"workload": {"nf4_m4096": "Linear4bit-style fused NF4 dequant+matmul, weight 4096x4096 bf16-origin bs64, M=4096 rows per GPU",
                                  "nf4dq_ffn": "fused NF4 + double-quant absmax, weight 11008x4096 bf16 bs64, M=4096",
                                  "int8_4096": "rowwise INT8 matmul_int8 4096x4096x4096 on int8 MFMA",
                                  "nf4_m1": "fused NF4 dequant+GEMV, weight 4096x4096 fp16 bs64, M=1, rotating over 64 layers"}[wl],
                     "global_rows": M_global, "rows_per_gpu": M, "N": N, "K": K,
                     "parallelism": f"rows sharded x{world}, weights replicated" + ((", all-gather of outputs (RCCL, blocking)" if args.sync_gather else ", all-gather of outputs (RCCL, async: overlaps the next step's GEMM)") if gathered is not None else ""),
                     "kernel": kernel_name}
    if wl == "nf4_m1":
        gbs = bytes_per_launch / (kern_ms * 1e-3) / 1e9
        out["metric"] = "effective GB/s, fused NF4 dequant+GEMV 4096x4096 M=1 (HBM, 64 rotating layers)"
        out["value"] = round(bytes_per_launch * 64 * world / (elapsed / args.steps) / 1e9, 2)
        out["unit"] = "GB/s"
        out["roofline"] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                           "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None,
                           "kernel_us": round(kern_ms * 1e3, 3)}
    else:
        tflops = total_flops / (elapsed / args.steps) / 1e12
        peak = PEAK_TFLOPS[name]
        kern_tflops = flops_per_step / (kern_ms * 1e-3) / 1e12
        out["metric"] = ("effective bf16 TFLOPS, fused NF4 dequant+matmul 4096x4096 @ M=4096" if wl == "nf4_m4096"
                         else f"effective TFLOPS, {wl}")
        out["value"] = round(tflops, 2)
        out["unit"] = "TFLOP/s"
        if gemm_only is not None:
            out["gemm_only"] = {"value": round(total_flops / (gemm_only / args.steps) / 1e12, 2), "unit": "TFLOP/s",
                                "ms_per_step": round(gemm_only / args.steps * 1e3, 5),
                                "note": "same steps without the output all-gather (communication-free scaling of the sharded GEMM)"}
        out["roofline"] = {"bound": "mfma", "achieved": round(kern_tflops, 2), "peak": peak, "unit": "TFLOP/s",
                           "frac": round(kern_tflops / peak, 4), "traffic": None,
                           "kernel_us": round(kern_ms * 1e3, 2)}

    if rank == 0 and wl == "nf4_m4096":
        if not args.no_gemv:
            # the M = 1 half of the metric: HBM-bound decode shape, 64 rotating layers
            layers = []
            for i in range(64):
                Wi = torch.randn(N, K, generator=g, device=dev, dtype=torch.float32).to(dt)
                layers.append(bnb.quantize_nf4(Wi, blocksize=64))
                del Wi
            x1 = torch.randn(1, K, generator=g, device=dev, dtype=torch.float32).to(dt)

            outs = [torch.empty(1, N, dtype=dt, device=dev) for _ in layers]

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
            for _ in range(3):
                graph.replay()
            torch.cuda.synchronize()
            us = event_time_ms(graph.replay, 20) / 64 * 1e3
            del outs
            nbytes = N * K // 2 + N * (K // 64) * 4 + K * 2 + N * 2
            gbs = nbytes / (us * 1e-6) / 1e9
            out["gemv"] = {"workload": "fused NF4 dequant+GEMV 4096x4096 M=1 bf16, 64 rotating layers (605 MB), one HIP graph of 64 launches (per-layer time includes the ~1 us launch boundary)",
                           "kernel": _native.last_kernel(), "us_per_layer": round(us, 3), "bytes_per_layer": nbytes,
                           "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                        "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None}}
            del layers
        if not args.no_gemv:
            # batch sizes between the two metric shapes (the reference's native path serves M <= 512): one HIP graph of
            # 8 calls per M, same weight; which kernel served it is recorded next to the time
            sweep = []
            for Ms in (2, 16, 64, 256, 1024):
                xs_ = torch.randn(Ms, K, generator=g, device=dev, dtype=torch.float32).to(dt)
                gr = torch.cuda.CUDAGraph()
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    bnb.matmul_4bit(xs_, packed, state)
                    with torch.cuda.graph(gr, stream=side):
                        for _ in range(8):
                            bnb.matmul_4bit(xs_, packed, state)
                torch.cuda.current_stream().wait_stream(side)
                for _ in range(3):
                    gr.replay()
                torch.cuda.synchronize()
                us = event_time_ms(gr.replay, 10) / 8 * 1e3
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
                out["roofline"]["empirical"] = {
                    "vendor_blas_bf16_same_shape_tflops": round(2.0 * M * N * K / (blas_ms * 1e-3) / 1e12, 1),
                    "vendor_blas_us": round(blas_ms * 1e3, 1),
                    "dtod_copy_gbs_read_plus_write": round(2.0 * (1 << 30) / (copy_ms * 1e-3) / 1e9, 0)}
            except Exception as e:  # context only: never fails the bench
                out["roofline"]["empirical"] = {"error": str(e)[:200]}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, M, N, K, 64, compress, dt)
    if rank == 0 and wl == "nf4_m4096":
        prof = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(prof):
            try:
                out["roofline"]["traffic"] = json.load(open(prof)).get("k_gemm256p_bytes_per_launch")
            except Exception:
                pass
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
