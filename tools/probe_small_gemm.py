"""Host-side cost of small GEMMs whose row count changes on every call (the training path at n_batch=20)."""
import time, sys
import torch
import torch.nn.functional as F

def run(tag):
    W = torch.randn(144, 48, device="cuda"); W2 = torch.randn(48, 48, device="cuda")
    sizes = [1000 + 37 * i for i in range(300)]
    xs = [torch.randn(n, 48, device="cuda") for n in sizes]
    for x in xs[:20]: F.linear(x, W)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for x in xs: F.linear(x, W); F.linear(x, W2)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%s: varying shapes: %.1f us per linear call (host), %.1f us incl. drain" % (tag, (t1 - t0) / 600 * 1e6, (t2 - t0) / 600 * 1e6))
    x = xs[0]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): F.linear(x, W); F.linear(x, W2)
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print("%s: fixed shape:    %.1f us per linear call (host)" % (tag, (t1 - t0) / 600 * 1e6))

print("default preferred blas:", torch.backends.cuda.preferred_blas_library())
run("default")
for b in ("cublas", "cublaslt"):
    try:
        torch.backends.cuda.preferred_blas_library(b)
        run(b)
    except Exception as e:
        print(b, "failed:", e)
