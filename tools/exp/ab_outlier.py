"""OutlierAwareLinear.forward of two library builds (prev = tools/exp/libmbnb_prev.so, new = the in-tree library) on one box: HIP graph of 20 calls,
4096^3 fp16 / bf16 with 16 and 40 outlier columns, output checksum.  Each library runs in its own subprocess."""
import os, sys, statistics, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = [(4096, 4096, 4096, "float16", 16), (4096, 4096, 4096, "bfloat16", 40), (2048, 4096, 4096, "float16", 16), (4096, 11008, 4096, "float16", 16)]


def child(path):
    sys.path.insert(0, ROOT)
    import torch
    from mps_bitsandbytes_amd import _native
    _native.LIB_PATH = path
    import mps_bitsandbytes_amd as bnb
    from mps_bitsandbytes_amd import synthetic
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
    for (M, N, K, dts, n_out) in CASES:
        dt = getattr(torch, dts)
        W = synthetic.normal((N, K), torch.float32, seed=1234, std=0.05)
        oidx = torch.arange(100, 100 + 90 * n_out, 90)
        W[:, oidx] *= 30.0
        torch.manual_seed(7)      # the bias of nn.Linear is drawn from the global generator
        lin = torch.nn.Linear(K, N, bias=True)
        lin.weight.data.copy_(W)
        oa = bnb.OutlierAwareLinear.from_linear(lin.to(dt).to(dev))
        x = synthetic.normal((M, K), dt, seed=4321).to(dev)
        y = oa(x)
        out[f"{M}x{N}x{K} {dts} {int(oa.outlier_indices.numel())} cols"] = (round(graph_us(lambda: oa(x)), 2), _native.last_kernel(), int(y.view(torch.int16).long().sum()))
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--child":
        child(sys.argv[2]); sys.exit(0)
    libs = [["prev", os.path.join(ROOT, "tools/exp/libmbnb_prev.so")], ["new", os.path.join(ROOT, "mps_bitsandbytes_amd/libmbnb_hip.so")]]
    res = {}
    for label, path in libs:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", path], capture_output=True, text=True, timeout=400)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(label, "FAILED", r.stdout[-2000:], r.stderr[-2000:]); sys.exit(1)
        res[label] = json.loads(line[0][7:])
    for k in res["prev"]:
        print(f"{k:40s} prev {res['prev'][k][0]:8.2f} us  new {res['new'][k][0]:8.2f} us   {res['new'][k][1]:20s} same bits: {res['prev'][k][2] == res['new'][k][2]}", flush=True)
