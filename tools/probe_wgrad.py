"""Weight-gradient GEMMs of the dense training path: dW[m,n] = G[N,m]^T X[N,n] with N in the millions.
Plain matmul (what autograd of F.linear issues) against the row-chunked batched form."""
import sys, time
import torch

def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_500_000
for m, n in ((192, 64), (64, 64), (8, 64), (384, 128)):
    G = torch.randn(N, m, device="cuda"); X = torch.randn(N, n, device="cuda")
    ref = G.t() @ X
    print("m=%d n=%d N=%d  plain %.3f ms" % (m, n, N, t(lambda: G.t() @ X)))
    for S in (64, 256, 1024, 4096):
        c = N // S
        def f():
            out = torch.bmm(G[:S * c].view(S, c, m).transpose(1, 2), X[:S * c].view(S, c, n)).sum(0)
            if S * c < N: out = out + G[S * c:].t() @ X[S * c:]
            return out
        err = (f() - ref).abs().max().item() / ref.abs().max().item()
        print("   chunks %5d: %.3f ms  (rel err %.1e)" % (S, t(f), err))
    print("   colsum: %.3f ms" % t(lambda: G.sum(0)))
