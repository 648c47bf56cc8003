# generates coissue.hip: how many VALU / transcendental fillers hide in an MFMA gap on gfx950 (one or two waves per SIMD)
import itertools
variants = []
def body(mfma, nfill, kind):
    # 4 independent accumulators; after each MFMA nfill fillers on independent registers
    lines = []
    fi = 0
    for j in range(4):
        if mfma == 32:
            lines.append(f"v_mfma_f32_32x32x16_bf16 %{j}, %4, %5, %{j}")
        elif mfma == 16:
            lines.append(f"v_mfma_f32_16x16x32_bf16 %{j}, %4, %5, %{j}")
        for f in range(nfill):
            r = 6 + (fi % 16)
            k = kind
            if kind == "mix":   # 1 exp in 4
                k = "exp" if fi % 4 == 0 else "fma"
            if kind == "mix2":  # 1 exp + 1 cvt + fma/max/add
                k = ["exp", "fma", "add", "max3", "cvt"][fi % 5]
            if k == "fma":
                lines.append(f"v_fma_f32 %{r}, %{r}, %22, %23")
            elif k == "add":
                lines.append(f"v_add_f32 %{r}, %{r}, %22")
            elif k == "exp":
                lines.append(f"v_exp_f32 %{r}, %{r}")
            elif k == "max3":
                lines.append(f"v_max3_f32 %{r}, %{r}, %22, %23")
            elif k == "cvt":
                lines.append(f"v_cvt_pk_bf16_f32 %{r}, %{r}, %22")
            elif k == "pkfma":
                lines.append(f"v_pk_fma_f32 %{24 + (fi % 4)}, %{24 + (fi % 4)}, %28, %28")
            fi += 1
    return "\\n\\t".join(lines)

src = ['#include <hip/hip_runtime.h>', '#include <cstdio>',
       'typedef float f32x4 __attribute__((ext_vector_type(4)));',
       'typedef float f32x2 __attribute__((ext_vector_type(2)));',
       'typedef float f32x16 __attribute__((ext_vector_type(16)));',
       'typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));']
cases = []
for mfma in (32, 16, 0):
    for kind in ("fma", "exp", "mix", "mix2", "cvt", "max3", "pkfma"):
        for nfill in ((0, 2, 4, 5, 6, 8, 12) if mfma == 32 else (0, 1, 2, 3, 4, 6) if mfma == 16 else (4,)):
            if nfill == 0 and kind != "fma":
                continue
            name = f"k_m{mfma}_{kind}_{nfill}"
            cases.append((name, mfma, kind, nfill))
            acc = "f32x16" if mfma == 32 else "f32x4"
            nacc = 16 if mfma == 32 else 4
            src.append(f'''
__global__ __launch_bounds__(256) void {name}(float* sink, int iters, unsigned long long* out) {{
  bf16x8 a, b;
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int i = 0; i < 8; ++i) {{
    s = s * 1664525u + 1013904223u; a[i] = (__bf16)(((int)(s >> 9) % 2001 - 1000) * 1e-3f);
    s = s * 1664525u + 1013904223u; b[i] = (__bf16)(((int)(s >> 9) % 2001 - 1000) * 1e-3f);
  }}
  {acc} c0, c1, c2, c3;
  for (int e = 0; e < {nacc}; ++e) {{ c0[e] = 0.f; c1[e] = 0.f; c2[e] = 0.f; c3[e] = 0.f; }}
  float v[16];
  for (int e = 0; e < 16; ++e) v[e] = -0.001f * (threadIdx.x + e);
  float k1 = 0.999f, k2 = -1e-4f;
  f32x2 p0 = {{0.1f, 0.2f}}, p1 = {{0.3f, 0.4f}}, p2 = {{0.5f, 0.6f}}, p3 = {{0.7f, 0.8f}}, pk = {{0.999f, 0.999f}};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {{
    asm volatile("{body(mfma, nfill, kind)}"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                 : "v"(a), "v"(b), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]),
                   "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]), "v"(k1), "v"(k2),
                   "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(pk));
  }}
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float acc = c0[0] + c1[0] + c2[0] + c3[0];
  for (int e = 0; e < 16; ++e) acc += v[e];
  acc += p0[0] + p1[0] + p2[0] + p3[0];
  if (acc == 12345.f) sink[0] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}}''')
# NB the fillers write registers declared as inputs: fine for a throughput probe (values are never checked)
src.append('''
template <typename K>
void run(const char* name, K kern, int wps, int per_iter_mfma, int per_iter_fill) {
  unsigned long long* d; float* sink; hipMalloc(&d, 64); hipMalloc(&sink, 64);
  unsigned long long h = 0;
  const int iters = 20000;
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    kern<<<256 * wps, 256>>>(sink, iters, d);
    hipEventRecord(e1); hipDeviceSynchronize(); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    hipEventElapsedTime(&ms, e0, e1);
  }
  double cyc = (double)h / iters;
  printf("%-22s wps=%d  %8.1f cyc/iter(4 mfma)  %6.1f cyc/mfma-gap  %6.2f cyc/filler  %.2f ms\\n", name, wps, cyc, per_iter_mfma ? cyc / per_iter_mfma : 0.0,
         per_iter_fill ? cyc / per_iter_fill : 0.0, ms);
  hipFree(d); hipFree(sink);
}
int main() {''')
for name, mfma, kind, nfill in cases:
    for wps in (1, 2):
        src.append(f'  run("{name}", {name}, {wps}, {4 if mfma else 0}, {4 * nfill});')
src.append('  return 0;\n}')
open("scratch/r3/coissue.hip", "w").write("\n".join(src))
