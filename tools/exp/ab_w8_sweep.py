"""W8A16 / FP8 linears at large M (dequantise + dense GEMM) of two library builds on the same box (argv: label=path ...; default prev = tools/exp/libmbnb_prev.so, new = the
in-tree library): device time per call from a HIP graph of 20 calls, the kernel that served it, and max |difference| against an f32 matmul of
the dequantised weight.  Each library runs in its own subprocess (one process loads one libmbnb_hip.so)."""
import os, sys, statistics, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = []
W8 = [(600, 4096, 4096), (1024, 4096, 4096), (2048, 4096, 4096), (4096, 4096, 4096), (1024, 11008, 4096), (4096, 2048, 2048)]


def child(path):
    sys.path.insert(0, ROOT)
    import torch
    from mps_bitsandbytes_amd import _native
    _native.LIB_PATH = path
    import mps_bitsandbytes_amd as bnb
    dev = torch.device("cuda:0")

    def graph_us(fn, n=20, reps=7):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for _ in range(n):
                    fn()
            ts = []
            for _ in range(reps):
                g.replay()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s); g.replay(); e1.record(s); e1.synchronize()
                ts.append(e0.elapsed_time(e1) / n * 1e3)
        return statistics.median(ts)

    out = {}
    g = torch.Generator(device=dev); g.manual_seed(11)
    for (M, N, K) in SHAPES:
        W = torch.randn(N, K, generator=g, device=dev).to(torch.bfloat16); x = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
        packed, st = bnb.quantize_nf4(W, blocksize=64)
        y = bnb.matmul_4bit(x, packed, st)
        kern = _native.last_kernel() if hasattr(_native, "last_kernel") else "?"
        ref = x.float() @ bnb.dequantize_nf4(packed, st).float().t()
        err = float((y.float() - ref).abs().max() / ref.abs().max())
        out[f"{M}x{N}x{K}"] = (round(graph_us(lambda: bnb.matmul_4bit(x, packed, st)), 2), kern, err)
    for (M, N, K) in W8:
        Wf = torch.randn(N, K, generator=g, device=dev).to(torch.bfloat16); x = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
        q, sc = bnb.quantize_rowwise(Wf)
        y = bnb.linear_int8(x, q, sc)
        kern = _native.last_kernel()
        ref = x.float() @ bnb.dequantize_rowwise(q, sc).float().t()
        err = float((y.float() - ref).abs().max() / ref.abs().max())
        out[f"w8a16 {M}x{N}x{K}"] = (round(graph_us(lambda: bnb.linear_int8(x, q, sc)), 2), kern + f" {int(y.view(torch.int16).long().sum()) & 0xFFFF:04x}", err)
        q8, s8 = bnb.quantize_fp8_e4m3(Wf)
        y = bnb.matmul_fp8_e4m3(x, q8, s8)
        kern = _native.last_kernel()
        out[f"fp8 {M}x{N}x{K}"] = (round(graph_us(lambda: bnb.matmul_fp8_e4m3(x, q8, s8)), 2), kern + f" {int(y.view(torch.int16).long().sum()) & 0xFFFF:04x}", 0.0)
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--child":
        child(sys.argv[2]); sys.exit(0)
    libs = [a.split("=", 1) for a in sys.argv[1:]] or [["prev", os.path.join(ROOT, "tools/exp/libmbnb_prev.so")], ["new", os.path.join(ROOT, "mps_bitsandbytes_amd/libmbnb_hip.so")]]
    res = {}
    for label, path in libs:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", path], capture_output=True, text=True, timeout=400)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(label, "FAILED", r.stdout[-2000:], r.stderr[-2000:]); sys.exit(1)
        res[label] = json.loads(line[0][7:])
    labels = [l for l, _ in libs]
    print(f"{'M x N x K':>20s} " + " ".join(f"{l:>10s} us  {'kernel (+ checksum)':<28s} {'rel err':>8s}" for l in labels))
    keys = [f"{p} {M}x{N}x{K}" for (M, N, K) in W8 for p in ("w8a16", "fp8")]
    for k in keys:
        print(f"{k:>20s} " + " ".join(f"{res[l][k][0]:10.2f}     {res[l][k][1]:<28s} {res[l][k][2]:8.1e}" for l in labels), flush=True)
