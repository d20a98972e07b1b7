// Why do the wavefronts of one modexp launch differ in speed by ~10 % (profiles/r04/launch_tail_probe_r04.jsonl: exactly 4 claims per
// wavefront cost 93 ms per claim, the aggregate rate corresponds to 84)?  Every wavefront of a resident grid (2 per SIMD) runs the SAME
// number of engine-like squaring sub-steps (random 29-bit limbs, 36-column window) and records its own duration on the 100 MHz wall clock
// together with where it ran (XCC_ID, HW_ID: SE / CU / SIMD).  The host prints the distribution per XCD and per SIMD slot.
//   hipcc -O3 --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=200000 wave_speed_map.hip -o wave_speed_map && ./wave_speed_map
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
constexpr uint32_t MASK29 = 0x1FFFFFFFu;

struct WaveRec { uint64_t t0, t1; uint32_t hw_id, xcc_id; };

__global__ void __launch_bounds__(256, 2) k_work(uint32_t* out, WaveRec* rec, uint32_t seed, int iters) {
  extern __shared__ __align__(16) uint32_t lds[];
  constexpr int W = 36;
  uint64_t acc[W];
  uint32_t A[W], N[W];
  const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
#pragma unroll
  for (int i = 0; i < W; i++) { A[i] = mix(seed + 977u * i + 131071u * tid) & MASK29; N[i] = mix(seed * 7u + 31u * i + 8191u * tid) & MASK29; acc[i] = i; }
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = mix(seed + i) & MASK29;
  __syncthreads();
  const uint64_t t0 = wall_clock64();
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int s = 0; s < W; s++) {
      const uint4 b4 = *reinterpret_cast<const uint4*>(lds + (((it * 9 + (s >> 2)) * 4) & 4092));
      const uint32_t b = (s & 3) == 0 ? b4.x : (s & 3) == 1 ? b4.y : (s & 3) == 2 ? b4.z : b4.w;
#pragma unroll
      for (int j = 0; j < 18; j++) acc[(s + j) % W] += (uint64_t)A[(2 * j + (s & 1)) % W] * b;
      uint32_t q = (uint32_t)acc[s] & MASK29;
      q = (uint32_t)__builtin_amdgcn_mov_dpp((int)q, 0x00, 0xF, 0xF, false);
#pragma unroll
      for (int j = 0; j < W; j++) acc[(s + j) % W] += (uint64_t)N[j] * q;
      const uint32_t lo = (uint32_t)acc[s] & MASK29;
      acc[(s + 1) % W] += acc[s] >> 29;
      acc[s] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x101, 0xF, 0xF, false);
    }
  }
  const uint64_t t1 = wall_clock64();
  uint64_t r = 0;
#pragma unroll
  for (int i = 0; i < W; i++) r ^= acc[i];
  out[tid] = (uint32_t)r ^ (uint32_t)(r >> 32);
  if ((threadIdx.x & 63) == 0) {
    WaveRec w;
    w.t0 = t0; w.t1 = t1;
    w.hw_id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID
    w.xcc_id = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID
    rec[blockIdx.x * 4 + (threadIdx.x >> 6)] = w;
  }
}

int main(int argc, char** argv) {
  const double T = argc > 1 ? atof(argv[1]) : 1.5;
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int blocks = p.multiProcessorCount * 2, waves = blocks * 4;
  const size_t lds = 72 * 1024;
  CHECK(hipFuncSetAttribute((const void*)k_work, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  uint32_t* d_out; WaveRec* d_rec;
  CHECK(hipMalloc(&d_out, (size_t)blocks * 256 * 4)); CHECK(hipMalloc(&d_rec, (size_t)waves * sizeof(WaveRec)));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  int iters = 64; float ms = 0;
  std::vector<WaveRec> rec(waves);
  for (int pass = 0; pass < 3; pass++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_work, dim3(blocks), dim3(256), lds, 0, d_out, d_rec, 12345u, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipGetLastError());
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (pass == 0) iters = (int)(iters * T * 1e3 / ms) + 1;
  }
  CHECK(hipMemcpy(rec.data(), d_rec, (size_t)waves * sizeof(WaveRec), hipMemcpyDeviceToHost));
  std::vector<double> dur(waves);
  uint64_t tmin = ~0ull;
  for (auto& w : rec) tmin = std::min(tmin, w.t0);
  for (int i = 0; i < waves; i++) dur[i] = (rec[i].t1 - rec[i].t0) / 100.0;            // microseconds on the 100 MHz clock
  std::vector<double> sorted = dur; std::sort(sorted.begin(), sorted.end());
  const double med = sorted[waves / 2];
  printf("{\"waves\": %d, \"iters\": %d, \"kernel_ms\": %.2f, \"wave_ms_min\": %.2f, \"p10\": %.2f, \"median\": %.2f, \"p90\": %.2f, \"max\": %.2f, \"max_over_min\": %.4f, \"latest_start_us\": %.1f}\n",
         waves, iters, ms, sorted[0] / 1e3, sorted[waves / 10] / 1e3, med / 1e3, sorted[waves * 9 / 10] / 1e3, sorted[waves - 1] / 1e3, sorted[waves - 1] / sorted[0],
         (double)(std::max_element(rec.begin(), rec.end(), [](const WaveRec& a, const WaveRec& b) { return a.t0 < b.t0; })->t0 - tmin) / 100.0);
  std::map<uint32_t, std::vector<double>> by_xcc, by_simd, by_se;
  for (int i = 0; i < waves; i++) {
    by_xcc[rec[i].xcc_id & 15].push_back(dur[i]);
    by_simd[(rec[i].hw_id >> 4) & 3].push_back(dur[i]);
    by_se[(rec[i].hw_id >> 13) & 7].push_back(dur[i]);
  }
  auto dump = [&](const char* name, std::map<uint32_t, std::vector<double>>& m) {
    for (auto& kv : m) {
      auto v = kv.second; std::sort(v.begin(), v.end());
      double sum = 0; for (double x : v) sum += x;
      printf("{\"group\": \"%s\", \"id\": %u, \"waves\": %zu, \"mean_ms\": %.3f, \"min_ms\": %.3f, \"max_ms\": %.3f, \"mean_over_overall_median\": %.4f}\n", name, kv.first, v.size(), sum / v.size() / 1e3,
             v.front() / 1e3, v.back() / 1e3, sum / v.size() / med);
    }
  };
  dump("xcc", by_xcc); dump("simd", by_simd); dump("se", by_se);
  return 0;
}
