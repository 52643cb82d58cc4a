// Issue cost (cycles per wave instruction) of the fp64 VALU operations the QP kernel is made of, on gfx950:
// 8 independent chains per op, 1 wave per SIMD and 4 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 64
template <int OP> __global__ void k(double* out, long long* cyc, double seed) {
  double a[8];
  for (int i = 0; i < 8; i++) a[i] = seed + i + threadIdx.x * 1e-3;
  double b = seed * 0.5 + 1.0, c = 1e-9;
  const int sel = threadIdx.x & 1;
  long long t0 = clock64();
#pragma unroll 1
  for (int r = 0; r < REP; r++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 3) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
      if (OP == 4) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[i]));
      if (OP == 5) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 6) { int lo = __double2loint(a[i]), hi = __double2hiint(a[i]); asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(sel), "v"(sel) : "vcc"); a[i] = __hiloint2double(hi, lo); }
      if (OP == 7) { int lo = __double2loint(a[i]); int s; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(lo)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "s"(s)); a[i] = __hiloint2double(__double2hiint(a[i]), lo); }
      if (OP == 8) { float f = (float)threadIdx.x + i; asm volatile("v_rcp_f32 %0, %0" : "+v"(f)); a[i] += f; }
      if (OP == 9) { int lo = __double2loint(a[i]); asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(lo)); a[i] = __hiloint2double(__double2hiint(a[i]), lo); }
      if (OP == 10) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(*(float*)&a[i]) : "v"((float)b), "v"((float)c));
      if (OP == 11) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    }
  }
  long long t1 = clock64();
  double s = 0; for (int i = 0; i < 8; i++) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int OP> void run(const char* name, int per_instr) {
  double* out; long long* cyc; hipMalloc(&out, 8 * 1024 * 1024); hipMalloc(&cyc, 1024 * 1024);
  for (int waves : {1, 2, 4, 8}) {   // waves per SIMD: a block of 4 * waves wavefronts on one CU
    const int bs = 256, nb = waves;   // nb blocks of 4 waves; launch 256*... enough blocks to fill: use 1 CU worth: nb blocks only (they land on different CUs) -> use one block of up to 1024 threads
    (void)bs; (void)nb;
    int threads = 64 * 4 * waves; if (threads > 1024) threads = 1024;
    k<OP><<<1, threads>>>(out, cyc, 1.5); hipDeviceSynchronize();
    k<OP><<<1, threads>>>(out, cyc, 1.5); hipDeviceSynchronize();
    std::vector<long long> h(threads / 64); hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    long long mx = 0; for (auto v : h) mx = v > mx ? v : mx;
    printf("%-14s waves/SIMD %d: %.2f clock64 ticks per wave-instruction (per SIMD: %.2f)\n", name, threads / 256, (double)mx / (REP * 8 * per_instr), (double)mx / (REP * 8 * per_instr) / (threads / 256));
  }
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0>("v_fma_f64", 1); run<1>("v_mul_f64", 1); run<2>("v_add_f64", 1); run<3>("v_rcp_f64", 1); run<4>("v_rsq_f64", 1); run<5>("v_max_f64", 1);
  run<6>("2x v_cndmask", 2); run<7>("readlane+add", 2); run<8>("v_rcp_f32(+cvt,add)", 1); run<9>("v_mov_dpp", 1); run<10>("v_fma_f32", 1); run<11>("v_pk_fma_f32", 1);
  // clock64 rate against wall time
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double* out; long long* cyc; hipMalloc(&out, 8 * 1024 * 1024); hipMalloc(&cyc, 1024 * 1024);
  hipEventRecord(e0); for (int i = 0; i < 200; i++) k<0><<<1, 256>>>(out, cyc, 1.5); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("one launch of %d fma: %lld ticks; 200 launches %.3f ms\n", REP * 8, c, ms);
  return 0;
}
