"""ab/libmippo_grutrace.so: the working tree with the GRU training forward's time loop (the
4-rows-per-workgroup form) stamped with s_memtime at its phase boundaries.  The product source
carries no instrumentation: this script patches a COPY of csrc/gru_mfma.hip and links it with
the tree's other objects.  Read with tools/trace_gru.py (MIPPO_LIB=ab/libmippo_grutrace.so).
Workgroup 7, wave 0 stamps steps 8..23 and leaves them in its own 4 rows of h_final."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
src = (ROOT / "nnx_ppo_amd/csrc/gru_mfma.hip").read_text()
k0 = src.index("gru_fwd_mfma_kernel(const float* __restrict__ gi")
k1 = src.index("// BPTT (formulas in gru.hip)")
body = src[k0:k1]
s0 = body.index("static_assert(PACK == 4")
s1 = body.index("\n  } else {\n", s0) + 1
sp = body[s0:s1]


def rep(old, new):
    global sp
    assert sp.count(old) == 1, (sp.count(old), old[:60])
    sp = sp.replace(old, new)


STAMP = ("{ __builtin_amdgcn_sched_barrier(0); if (tid == 0 && t >= 8 && t < 24) "
         "trc[((int)t - 8) * 8 + (%d)] = __builtin_amdgcn_s_memtime(); "
         "__builtin_amdgcn_sched_barrier(0); }\n")
rep("    auto step = [&](int64_t t, Slot& gcur) {\n",
    "    __shared__ unsigned long long trc[16 * 8];\n"
    "    auto step = [&](int64_t t, Slot& gcur) {\n" + STAMP % 0)
rep("      f32x4 acc[UTW][3];\n", STAMP % 1 + "      f32x4 acc[UTW][3];\n")            # proj done
rep("      unsigned char* const rc0 = pk.rec((int)(t & 1), rr);\n",
    STAMP % 2 + "      unsigned char* const rc0 = pk.rec((int)(t & 1), rr);\n")      # h MFMAs issued
rep("        unsigned char* const rc = rc0 + 4 * ucol[ui];\n",
    "        { volatile float sink = hnew + r + z + n + qn; (void)sink; }\n" + STAMP % 3 +
    "        unsigned char* const rc = rc0 + 4 * ucol[ui];\n")                        # gate math done
rep("      if (t != 0) pk.sweep((int)((t - 1) & 1), tid);\n",
    "      if (t != 0) pk.sweep((int)((t - 1) & 1), tid);\n" + STAMP % 7)     # sweep issued
rep("      load_step(t + PFW, gcur);\n      __syncthreads();\n",
    STAMP % 4 + "      load_step(t + PFW, gcur);\n" + STAMP % 5 + "      __syncthreads();\n" +
    STAMP % 6)
rep("      if (svalid) h_final[srowc * (unsigned)H + ucol[ui]] = hc[ui];\n    }\n",
    "      if (svalid) h_final[srowc * (unsigned)H + ucol[ui]] = hc[ui];\n    }\n"
    "    __syncthreads();\n"
    "    if (blockIdx.x == 7 && tid < 64) {\n"
    "      unsigned long long* o = reinterpret_cast<unsigned long long*>(h_final + row0 * H);\n"
    "      for (int i = tid; i < 16 * 8; i += 64) o[i] = trc[i];\n"
    "    }\n")
out = Path("/tmp/exp/gru_trace.hip")
out.parent.mkdir(exist_ok=True)
out.write_text(src[:k0] + body[:s0] + sp + body[s1:] + src[k1:])
subprocess.check_call([sys.executable, "-m", "nnx_ppo_amd.csrc.build"], cwd=ROOT,
                      stdout=subprocess.DEVNULL)
hipcc = "/opt/rocm/bin/hipcc"
subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                       f"-I{ROOT}/include", f"-I{ROOT}/nnx_ppo_amd/csrc", "-c", str(out),
                       "-o", "/tmp/exp/gru_trace.o"])
objs = [str(p) for p in (ROOT / "nnx_ppo_amd/csrc/build").glob("*.o") if p.name != "gru_mfma.o"]
(ROOT / "ab").mkdir(exist_ok=True)
subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                       str(ROOT / "ab/libmippo_grutrace.so"), *objs, "/tmp/exp/gru_trace.o"])
print("ab/libmippo_grutrace.so")
