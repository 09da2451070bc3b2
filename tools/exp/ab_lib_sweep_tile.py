"""Large M (decode-once path): matmul_4bit of two library builds on the same box (argv: label=path ...; default prev = tools/exp/libmbnb_prev.so, new = the
in-tree library): device time per call from a HIP graph of 20 calls, the kernel that served it, and max |difference| against an f32 matmul of
the dequantised weight.  Each library runs in its own subprocess (one process loads one libmbnb_hip.so)."""
import os, sys, statistics, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(200, 1000, 960), (65, 256, 896), (128, 4096, 1088), (300, 2048, 1600), (512, 4096, 4160), (96, 1024, 4032), (640, 1024, 2048), (256, 520, 3008), (400, 777, 1792), (64, 4096, 448), (350, 3000, 1344), (1000, 512, 4096)]


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
        csum = int(y.view(torch.int16).long().sum())
        packed2, st2 = bnb.quantize_nf4(W, blocksize=64, compress_statistics=True)
        y2 = bnb.matmul_4bit(x, packed2, st2)
        csum2 = int(y2.view(torch.int16).long().sum())
        out[f"{M}x{N}x{K}"] = (round(graph_us(lambda: bnb.matmul_4bit(x, packed, st)), 2), kern, err, csum, round(graph_us(lambda: bnb.matmul_4bit(x, packed2, st2)), 2), csum2)
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
    print(f"{'M x N x K':>20s} " + " ".join(f"{l:>10s} us  {'kernel':<20s} {'rel err':>8s} {'nested us':>10s}" for l in labels) + "  same bits (plain / nested absmax)")
    for (M, N, K) in SHAPES:
        k = f"{M}x{N}x{K}"
        same = (len({res[l][k][3] for l in labels}) == 1, len({res[l][k][5] for l in labels}) == 1)
        print(f"{k:>20s} " + " ".join(f"{res[l][k][0]:10.2f}     {res[l][k][1]:<20s} {res[l][k][2]:8.1e} {res[l][k][4]:10.2f}" for l in labels) + f"  {same[0]} / {same[1]}", flush=True)
