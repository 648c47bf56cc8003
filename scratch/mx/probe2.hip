// Layout discovery for v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 x fp8) on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// raw per-lane operands: Araw/Braw [64 lanes][32 bytes], sA/sB [64] dwords
__global__ void k(const uint8_t* Araw, const uint8_t* Braw, const unsigned* sA, const unsigned* sB, float* Craw) {
  const int l = threadIdx.x;
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = *reinterpret_cast<const int*>(Araw + l * 32 + i * 4);
    b[i] = *reinterpret_cast<const int*>(Braw + l * 32 + i * 4);
  }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, (int)sA[l], 0, (int)sB[l]);
  for (int i = 0; i < 4; ++i) Craw[l * 4 + i] = c[i];
}
static float e4m3(uint8_t v) {
  int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = (e == 0) ? ldexpf((float)m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
  return s ? -x : x;
}
int main() {
  uint8_t *dA, *dB; unsigned *dsa, *dsb; float* dC;
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dC, 1024);
  std::vector<uint8_t> A(2048), B(2048); std::vector<unsigned> sa(64, 127), sb(64, 127); std::vector<float> C(256);
  auto run = [&]() {
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
    hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dC);
    hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
  };
  // 1) which (lane, byte) of A pairs with which (lane, byte) of B?  A = single 1.0 at (la, pa); B: every (lane, byte) holds a code
  //    unique within its column group: run twice with codes (byte index) and (lane group)
  printf("A(lane,byte) -> matching B (group,byte) for output col 0 and the output element that is non-zero\n");
  for (int la : {0, 16, 32, 48, 5}) for (int pa : {0, 1, 3, 4, 15, 16, 31}) {
    std::fill(A.begin(), A.end(), 0); A[la * 32 + pa] = 0x38;  // 1.0
    for (int l = 0; l < 64; ++l) for (int p = 0; p < 32; ++p) B[l * 32 + p] = 0x20 + p;      // value encodes byte
    run();
    int nz = -1; float v1 = 0;
    for (int i = 0; i < 256; ++i) if (C[i] != 0.f && (i / 4) % 16 == 0) { nz = i; v1 = C[i]; break; }
    for (int l = 0; l < 64; ++l) for (int p = 0; p < 32; ++p) B[l * 32 + p] = 0x20 + (l >> 4);  // value encodes lane group
    run();
    float v2 = nz >= 0 ? C[nz] : 0;
    int pb = -1, gb = -1;
    for (int p = 0; p < 32; ++p) if (e4m3(0x20 + p) == v1) pb = p;
    for (int g = 0; g < 4; ++g) if (e4m3(0x20 + g) == v2) gb = g;
    int cnt = 0; for (int i = 0; i < 256; ++i) cnt += C[i] != 0.f;
    printf("A lane %2d byte %2d -> B group %d byte %2d ; C lane %d reg %d nonzero (total nonzero %d)\n", la, pa, gb, pb, nz / 4, nz % 4, cnt);
  }
  // 2) scales: all data 1.0; bump one lane's A scale byte b to 128 (x2): which outputs change, by how much
  std::fill(A.begin(), A.end(), 0x38); std::fill(B.begin(), B.end(), 0x38);
  for (int ls : {0, 1, 16, 33}) for (int byte = 0; byte < 4; ++byte) {
    std::fill(sa.begin(), sa.end(), 0x7f7f7f7fu); std::fill(sb.begin(), sb.end(), 0x7f7f7f7fu);
    sa[ls] = (0x7f7f7f7fu & ~(0xffu << (8 * byte))) | (128u << (8 * byte));
    run();
    printf("A scale lane %2d byte %d = 2.0 :", ls, byte);
    int shown = 0;
    for (int i = 0; i < 256 && shown < 6; ++i) if (C[i] != 128.f) { printf(" C[lane %d reg %d]=%g", i / 4, i % 4, C[i]); ++shown; }
    int cnt = 0; for (int i = 0; i < 256; ++i) cnt += C[i] != 128.f;
    printf("  (changed %d)\n", cnt);
  }
  for (int ls : {0, 1, 16, 33}) {
    std::fill(sa.begin(), sa.end(), 0x7f7f7f7fu); std::fill(sb.begin(), sb.end(), 0x7f7f7f7fu);
    sb[ls] = 0x7f7f7f80u;
    run();
    printf("B scale lane %2d byte 0 = 2.0 :", ls);
    int shown = 0;
    for (int i = 0; i < 256 && shown < 6; ++i) if (C[i] != 128.f) { printf(" C[lane %d reg %d]=%g", i / 4, i % 4, C[i]); ++shown; }
    int cnt = 0; for (int i = 0; i < 256; ++i) cnt += C[i] != 128.f;
    printf("  (changed %d)\n", cnt);
  }
  // both: A scale (row 2, block 1) = 4, B scale (col 3, block 1) = 8, and B scale (col 3, block 2) = 2
  std::fill(sa.begin(), sa.end(), 0x7f7f7f7fu); std::fill(sb.begin(), sb.end(), 0x7f7f7f7fu);
  sa[16 + 2] = 0x7f7f7f81u; sb[16 + 3] = 0x7f7f7f82u; sb[32 + 3] = 0x7f7f7f80u;
  run();
  printf("combo: C[row2][col3] = %g (expect 32*(1+32+2+1)=1152), C[row2][col0] = %g (expect 32*(1+4+1+1)=224), C[row0][col3] = %g (expect 32*(1+8+2+1)=384)\n",
         C[3 * 4 + 2], C[0 * 4 + 2], C[3 * 4 + 0]);
  for (unsigned e : {120u, 124u, 125u, 126u, 127u, 128u, 129u, 131u}) {
    std::fill(sa.begin(), sa.end(), 0x7f7f7f00u | e); std::fill(sb.begin(), sb.end(), 0x7f7f7f7fu);
    run();
    printf("all A scales = %u: C[0] = %g (128 * 2^(e-127) = %g)\n", e, C[0], 128.0 * ldexp(1.0, (int)e - 127));
  }
  {
    std::fill(sa.begin(), sa.end(), 0x7f7f7f7fu); std::fill(sb.begin(), sb.end(), 0x7f7f7f7fu);
    for (int l = 0; l < 64; ++l) { sa[l] = 0x7f7f7f00u | (124 + (l * 7) % 6); sb[l] = 0x7f7f7f00u | (125 + (l * 5) % 5); }
    run();
    double worst = 0;
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
      double ref = 0;
      for (int g = 0; g < 4; ++g) ref += 32.0 * ldexp(1.0, (int)(sa[g * 16 + m] & 255) - 127) * ldexp(1.0, (int)(sb[g * 16 + n] & 255) - 127);
      double got = C[((m >> 2) * 16 + n) * 4 + (m & 3)];
      worst = fmax(worst, fabs(got - ref) / ref);
    }
    printf("mixed scales, unit data: worst rel err %g\n", worst);
  }
  // which scale lane-group covers which data (lane-group, byte)?  data = 1.0 only at (g, p) of A row 0 / B col 0
  printf("data (g,p) -> A-scale lane group that scales it:\n");
  for (int g = 0; g < 4; ++g) for (int p : {0, 7, 8, 15, 16, 23, 24, 31}) {
    std::fill(A.begin(), A.end(), 0); std::fill(B.begin(), B.end(), 0);
    A[(g * 16) * 32 + p] = 0x38; B[(g * 16) * 32 + p] = 0x38;
    int who = -1;
    for (int gs = 0; gs < 4; ++gs) {
      std::fill(sa.begin(), sa.end(), 0x7f7f7f7fu); std::fill(sb.begin(), sb.end(), 0x7f7f7f7fu);
      sa[gs * 16] = 0x7f7f7f80u;
      run();
      if (C[0] == 2.f) who = gs;
    }
    printf("  (g %d, byte %2d) -> scale group %d\n", g, p, who);
  }
  // 3) C layout: A row r has value (r+1) in all k of lane r (group 0 only), B = 1
  std::fill(sa.begin(), sa.end(), 0x7f7f7f7fu); std::fill(sb.begin(), sb.end(), 0x7f7f7f7fu);
  std::fill(A.begin(), A.end(), 0); std::fill(B.begin(), B.end(), 0);
  for (int r = 0; r < 16; ++r) A[r * 32] = 0x20 + r;   // distinct value per A lane r (byte 0, group 0)
  for (int c = 0; c < 16; ++c) B[c * 32] = 0x38;       // B lane c byte 0 = 1
  run();
  printf("C layout: lane l reg i holds A-lane:");
  for (int l : {0, 1, 15, 16, 17, 32, 48, 63}) { printf("  l%d:", l); for (int i = 0; i < 4; ++i) { int who = -1; for (int r = 0; r < 16; ++r) if (e4m3(0x20 + r) == C[l * 4 + i]) who = r; printf("%d,", who); } }
  printf("\n");
  return 0;
}
