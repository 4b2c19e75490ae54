"""Build red-gnn_amd/libredgnn_wpt.so: the library with cycle counters around the phases of layer_fwd_wp_kernel (item load / fill /
phase 1 / phase 2 / tickets), read back through rg_debug_wp_timing - what tools/probe_wp_phases.py runs on (RG_LIB=...).  The
instrumented source is generated from csrc/layer_fwd_wp.hip and never committed.    python tools/make_wp_timing_variant.py"""
import os, subprocess, sys, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(ROOT, "red-gnn_amd", "csrc")
s = open(os.path.join(C, "layer_fwd_wp.hip")).read()

def sub(old, new):
    global s
    assert old in s, old[:60]
    s = s.replace(old, new, 1)

sub("namespace rgwp {\nnamespace {",
    "__device__ unsigned long long rg_wp_tm[16];\nextern \"C\" int rg_debug_wp_timing(unsigned long long* out, int reset) {\n"
    "  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rg_wp_tm), sizeof(rg_wp_tm)) != hipSuccess) return 1;\n"
    "  if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(rg_wp_tm), z, sizeof(z)) != hipSuccess) return 1; }\n"
    "  return 0;\n}\nnamespace rgwp {\nnamespace {")
sub("  auto run_item = [&](long long item) {",
    "  unsigned long long tm[8] = {0,0,0,0,0,0,0,0};\n  unsigned long long t_prev = __builtin_readcyclecounter();\n"
    "  auto lap = [&](int k) { const unsigned long long t = __builtin_readcyclecounter(); tm[k] += t - t_prev; t_prev = t; };\n"
    "  auto run_item = [&](long long item) {\n    lap(0);")
sub("    const int row0 = A.pack[pi].x;\n    uint32_t todo = wave_or(w0 | w1);", "    const int row0 = A.pack[pi].x;\n    uint32_t todo = wave_or(w0 | w1);\n    lap(1);")
sub("      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, \"wavefront\");\n      __builtin_amdgcn_wave_barrier();\n      // ---- phase 1:",
    "      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, \"wavefront\");\n      __builtin_amdgcn_wave_barrier();\n      lap(2); tm[6] += qn; tm[7] += 1;\n      // ---- phase 1:")
sub("      // ---- phase 2: the lane groups take contiguous shares", "      lap(3);\n      // ---- phase 2: the lane groups take contiguous shares")
sub("      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, \"wavefront\");      // the queue is read out before the next fill overwrites it\n      __builtin_amdgcn_wave_barrier();",
    "      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, \"wavefront\");      // the queue is read out before the next fill overwrites it\n      __builtin_amdgcn_wave_barrier();\n      lap(4);")
sub("    for (int k = 0; k < cnt; ++k) run_item(first + k);\n  }\n}",
    "    for (int k = 0; k < cnt; ++k) run_item(first + k);\n  }\n  lap(0);\n"
    "  if (lane == 0) { for (int k = 0; k < 8; ++k) atomicAdd(&rg_wp_tm[k], tm[k]); atomicAdd(&rg_wp_tm[8], 1ull); }\n}")
tmp = os.path.join(C, "_wp_timing_tmp.hip")
open(tmp, "w").write(s)
try:
    obj = "/tmp/variant_wpt.o"
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-I" + os.path.join(ROOT, "include"), "-I" + C,
                           "-c", tmp, "-o", obj])
finally:
    os.remove(tmp)
objs = [o for o in glob.glob(os.path.join(C, "build", "*.o")) if not o.endswith("/layer_fwd_wp.o")]
out = os.path.join(ROOT, "red-gnn_amd", "libredgnn_wpt.so")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + [obj])
print("built", out)
