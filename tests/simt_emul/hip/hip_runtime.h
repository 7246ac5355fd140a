// TEST INFRASTRUCTURE ONLY -- a tiny host-side SIMT emulator that stands in for <hip/hip_runtime.h> so that the
// UNMODIFIED kernel sources under uglad_amd/csrc/ can be compiled with clang++ for x86 and executed on the CPU by
// tests/ (indexing, barrier structure, MFMA lane maps and reductions get checked without a GPU; the same build runs
// under -fsanitize=address,undefined).  The product never loads this; `uglad_amd` only ever dlopens the gfx950 build.
//
// Model: one workgroup at a time; every work-item is a ucontext fiber that runs until it reaches __syncthreads() or a
// wave-collective (shuffle, MFMA), where it yields to a round-robin scheduler.  Wave = 64 consecutive work-items.
// `__shared__` becomes function-local `static` storage (one workgroup runs at a time, so that is the workgroup's LDS).
// Not modelled: races between barriers (fibers run one after the other), LDS capacity, register pressure, timing.
#pragma once
#define UGLAD_SIMT_EMUL 1  /* host build on the SIMT emulator: no streams, no graphs */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;
typedef int hipError_t;
static const hipError_t hipSuccess = 0;
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, void*) { memset(p, v, n); return hipSuccess; }

inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }

namespace simt {

constexpr int kWave = 64;
constexpr size_t kStack = 256 * 1024;

struct Fiber {
  ucontext_t ctx;
  char* stack = nullptr;
  bool done = false;
  const volatile unsigned* wait_ptr = nullptr;
  unsigned wait_val = 0;
  unsigned n_coll = 0;  // wave-wide collectives executed (parity selects the exchange buffer)
  unsigned n_shfl = 0;  // shuffles executed since the last workgroup barrier
  unsigned shfl_epoch = 0xffffffffu;
  unsigned n_wsync = 0;    // explicit wave rendezvous (wave_sync_point) since the last workgroup barrier
  dim3 tid;
};

constexpr int kRing = 256;  // a lane can run ahead of a slow reader by a whole butterfly + a few broadcasts
struct WaveState {
  uint32_t slot[2][kWave];   // MFMA operand exchange (wave-wide collective)
  uint32_t slot2[2][kWave];
  uint64_t slot64a[2][kWave];  // the same for the fp64 MFMA
  uint64_t slot64b[2][kWave];
  uint32_t sh_val[kWave][kRing];  // shuffles: per-lane ring of (value, tag); a reader only waits for its SOURCE lane, so
  uint64_t sh_tag[kWave][kRing];
  unsigned sh_seq[kWave][kRing];  // bumped at every write of the slot: what a waiting reader watches  // lanes that sit out a divergent region (as on hardware) do not block the others
  int arrived = 0;
  unsigned gen = 0;
  int size = kWave;
};

struct State {
  std::vector<Fiber> fibers;
  std::vector<WaveState> waves;
  ucontext_t sched;
  int cur = -1;
  int nthreads = 0;
  int arrived = 0;
  unsigned barrier_gen = 0;
  unsigned long progress = 0;
  dim3 bid, bdim, gdim;
  std::function<void()> body;
};

inline State& st() {
  static State s;
  return s;
}

inline void yield_to_sched() {
  State& s = st();
  swapcontext(&s.fibers[s.cur].ctx, &s.sched);
}

inline void wait_on(const volatile unsigned* p, unsigned v) {
  State& s = st();
  Fiber& f = s.fibers[s.cur];
  f.wait_ptr = p;
  f.wait_val = v;
  yield_to_sched();
}

inline void syncthreads() {
  State& s = st();
  const unsigned gen = s.barrier_gen;
  if (++s.arrived == s.nthreads) {
    s.arrived = 0;
    ++s.barrier_gen;
    ++s.progress;
  } else {
    wait_on(&s.barrier_gen, gen);
  }
}

inline WaveState& my_wave() {
  State& s = st();
  return s.waves[s.cur / kWave];
}

inline void wave_sync() {
  State& s = st();
  WaveState& w = my_wave();
  const unsigned gen = w.gen;
  if (++w.arrived == w.size) {
    w.arrived = 0;
    ++w.gen;
    ++s.progress;
  } else {
    wait_on(&w.gen, gen);
  }
}

// A rendezvous of the lanes of one wave in wave-uniform code (UGLAD_WAVE_SYNC): like a workgroup barrier it is a convergence
// point, so the shuffle counts of lanes that drifted apart in a divergent region line up again behind it.
inline void wave_sync_point() {
  State& s = st();
  Fiber& f = s.fibers[s.cur];
  if (f.shfl_epoch != s.barrier_gen) {
    f.shfl_epoch = s.barrier_gen;
    f.n_wsync = 0;
  }
  wave_sync();
  ++f.n_wsync;
  f.n_shfl = 0;
}

// Lanes that take part in a shuffle must have executed the same number of shuffles before it (true for converged code and
// for lane groups that diverge together, e.g. the lane pairs of the secular solver).
template <class T>
inline T shfl_idx(T v, int src_lane) {
  static_assert(sizeof(T) == 4, "32-bit shuffles only");
  State& s = st();
  WaveState& w = my_wave();
  Fiber& f = s.fibers[s.cur];
  const int lane = s.cur % kWave;
  // tags restart at every workgroup barrier (a convergence point), so lanes whose shuffle counts drifted apart inside a
  // divergent region (different iteration counts per lane pair) line up again afterwards
  if (f.shfl_epoch != s.barrier_gen) {
    f.shfl_epoch = s.barrier_gen;
    f.n_shfl = 0;
    f.n_wsync = 0;
  }
  // (tags must grow monotonically per lane: 20 bits of shuffle count under 12 bits of wave rendezvous under the barrier count)
  const uint64_t tag = ((uint64_t)s.barrier_gen << 32) + ((uint64_t)f.n_wsync << 20) + (++f.n_shfl);
  const int slot = tag % kRing;
  memcpy(&w.sh_val[lane][slot], &v, 4);
  w.sh_tag[lane][slot] = tag;
  ++w.sh_seq[lane][slot];
  ++s.progress;
  const int src = src_lane & (kWave - 1);
  if (src >= w.size) return v;
  while (w.sh_tag[src][slot] != tag) {
    if (w.sh_tag[src][slot] > tag) {
      fprintf(stderr, "simt_emul: shuffle ring overrun (lane %d reading lane %d)\n", lane, src);
      abort();
    }
    wait_on(&w.sh_seq[src][slot], w.sh_seq[src][slot]);
  }
  T r;
  memcpy(&r, &w.sh_val[src][slot], 4);
  return r;
}

// DPP move (v_mov_b32_dpp): the controls the kernels use -- quad_perm (0x00-0xff), row_ror:n (0x121-0x12f), row_bcast:15
// (0x142), row_bcast:31 (0x143) -- with row/bank write masks.  A lane that is masked out, or has no source, keeps `old`.
inline int update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  State& s = st();
  const int lane = s.cur % kWave, row = lane / 16, bank = (lane % 16) / 4;
  int srcl = -1;
  if (ctrl <= 0xff) srcl = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
  else if (ctrl >= 0x121 && ctrl <= 0x12f) srcl = row * 16 + ((lane % 16 - (ctrl & 0xf) + 16) % 16);
  else if (ctrl == 0x140) srcl = row * 16 + (15 - lane % 16);          // row_mirror
  else if (ctrl == 0x141) srcl = (lane & ~7) | (7 - (lane & 7));          // row_half_mirror
  else if (ctrl == 0x142) srcl = (row >= 1) ? row * 16 - 1 : -1;
  else if (ctrl == 0x143) srcl = (lane >= 32) ? 31 : -1;
  else {
    fprintf(stderr, "simt_emul: unsupported dpp_ctrl 0x%x\n", ctrl);
    abort();
  }
  const int got = shfl_idx(src, srcl >= 0 ? srcl : lane);
  const bool enabled = ((row_mask >> row) & 1) && ((bank_mask >> bank) & 1);
  if (!enabled) return old;
  if (srcl < 0) return bound_ctrl ? 0 : old;
  return got;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
// v_mfma_f32_32x32x2_f32: lane l gives A[l&31][l>>5], B[l>>5][l&31]; C reg e of lane l = C[(e&3)+8(e>>2)+4(l>>5)][l&31];
// numerics = k-ordered fmaf chain.
inline f32x16 mfma_32x32x2f32(float a, float b, f32x16 c, int, int, int) {
  State& s = st();
  WaveState& w = my_wave();
  Fiber& f = s.fibers[s.cur];
  const int lane = s.cur % kWave, par = f.n_coll++ & 1;
  memcpy(&w.slot[par][lane], &a, 4);
  memcpy(&w.slot2[par][lane], &b, 4);
  wave_sync();
  const int col = lane & 31;
  for (int e = 0; e < 16; ++e) {
    const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
    float acc = c[e];
    for (int k = 0; k < 2; ++k) {
      float av, bv;
      memcpy(&av, &w.slot[par][k * 32 + row], 4);
      memcpy(&bv, &w.slot2[par][k * 32 + col], 4);
      acc = fmaf(av, bv, acc);
    }
    c[e] = acc;
  }
  return c;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// v_mfma_f32_16x16x4_f32: lane l gives A[l&15][l>>4], B[l>>4][l&15]; C reg r of lane l = C[4(l>>4)+r][l&15].
inline f32x4 mfma_16x16x4f32(float a, float b, f32x4 c, int, int, int) {
  State& s = st();
  WaveState& w = my_wave();
  Fiber& f = s.fibers[s.cur];
  const int lane = s.cur % kWave, par = f.n_coll++ & 1;
  memcpy(&w.slot[par][lane], &a, 4);
  memcpy(&w.slot2[par][lane], &b, 4);
  wave_sync();
  const int col = lane & 15;
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * (lane >> 4) + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) {
      float av, bv;
      memcpy(&av, &w.slot[par][k * 16 + row], 4);
      memcpy(&bv, &w.slot2[par][k * 16 + col], 4);
      acc = fmaf(av, bv, acc);
    }
    c[r] = acc;
  }
  return c;
}

typedef double f64x4_emul __attribute__((ext_vector_type(4)));
// v_mfma_f64_16x16x4_f64: lane l gives A[l&15][l>>4], B[l>>4][l&15]; C reg r of lane l = C[(l>>4) + 4r][l&15] (NOT the f32 row map).
inline f64x4_emul mfma_16x16x4f64(double a, double b, f64x4_emul c, int, int, int) {
  State& s = st();
  WaveState& w = my_wave();
  Fiber& f = s.fibers[s.cur];
  const int lane = s.cur % kWave, par = f.n_coll++ & 1;
  memcpy(&w.slot64a[par][lane], &a, 8);
  memcpy(&w.slot64b[par][lane], &b, 8);
  wave_sync();
  const int col = lane & 15;
  for (int r = 0; r < 4; ++r) {
    const int row = (lane >> 4) + 4 * r;
    double acc = c[r];
    for (int k = 0; k < 4; ++k) {
      double av, bv;
      memcpy(&av, &w.slot64a[par][k * 16 + row], 8);
      memcpy(&bv, &w.slot64b[par][k * 16 + col], 8);
      acc = fma(av, bv, acc);
    }
    c[r] = acc;
  }
  return c;
}

// 64-bit shuffle: two 32-bit ones (as the hardware does it)
inline double shfl_idx(double v, int src_lane) {
  uint32_t h[2];
  memcpy(h, &v, 8);
  h[0] = shfl_idx(h[0], src_lane);
  h[1] = shfl_idx(h[1], src_lane);
  memcpy(&v, h, 8);
  return v;
}

inline void fiber_entry() {
  State& s = st();
  s.body();
  s.fibers[s.cur].done = true;
  ++s.progress;
  swapcontext(&s.fibers[s.cur].ctx, &s.sched);
}

inline void run_block(dim3 grid, dim3 block, dim3 bidx, const std::function<void()>& body) {
  const unsigned bx = bidx.x;
  State& s = st();
  const int n = (int)(block.x * block.y * block.z);
  if ((int)s.fibers.size() < n) {
    const size_t old = s.fibers.size();
    s.fibers.resize(n);
    for (size_t i = old; i < (size_t)n; ++i) s.fibers[i].stack = (char*)malloc(kStack);
  }
  s.nthreads = n;
  s.arrived = 0;
  ++s.barrier_gen;  // fresh shuffle epoch for the new workgroup
  s.bid = bidx;
  s.bdim = block;
  s.gdim = grid;
  s.body = body;
  const int nw = (n + kWave - 1) / kWave;
  s.waves.assign(nw, WaveState());
  for (int wv = 0; wv < nw; ++wv) s.waves[wv].size = (wv == nw - 1) ? n - wv * kWave : kWave;
  for (int i = 0; i < n; ++i) {
    Fiber& f = s.fibers[i];
    f.done = false;
    f.wait_ptr = nullptr;
    f.n_coll = 0;
    f.n_shfl = 0;
    f.shfl_epoch = 0xffffffffu;
    f.tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = kStack;
    f.ctx.uc_link = &s.sched;
    makecontext(&f.ctx, (void (*)())fiber_entry, 0);
  }
  int remaining = n;
  // Scheduling policy (UGLAD_EMUL_SCHED): "fair" (default) gives every runnable fiber one turn per round, so the waves advance in
  // step; "ahead" / "behind" always resume the runnable fiber of the LOWEST / HIGHEST wave, so that wave runs as far ahead of the others
  // as its own collectives allow -- up to the next workgroup barrier.  That is the skew hardware produces and a fair schedule never
  // does: a missing __syncthreads() between two phases that reuse one LDS region shows up as wrong results under these two policies.
  static const int policy = [] {
    const char* e = getenv("UGLAD_EMUL_SCHED");
    return !e ? 0 : (e[0] == 'a' ? 1 : (e[0] == 'b' ? 2 : 0));
  }();
  while (remaining > 0) {
    const unsigned long before = s.progress;
    bool ran = false;
    bool ran_in_wave = false;
    for (int k = 0; k < n; ++k) {
      const int i = (policy == 2) ? n - 1 - k : k;
      if (policy != 0 && k > 0 && (k % kWave) == 0) {  // wave boundary: if the preferred wave made progress, give it the next turn too
        if (ran_in_wave) break;
      }
      Fiber& f = s.fibers[i];
      if (f.done) continue;
      if (f.wait_ptr && *f.wait_ptr == f.wait_val) continue;
      f.wait_ptr = nullptr;
      s.cur = i;
      ran = true;
      ran_in_wave = true;
      swapcontext(&s.sched, &f.ctx);
      if (f.done) --remaining;
    }
    if (!ran && s.progress == before && remaining > 0) {
      fprintf(stderr, "simt_emul: deadlock in block %u (%d work-items stuck at a barrier / wave collective)\n", bx, remaining);
      abort();
    }
  }
  s.cur = -1;
}

inline const dim3& tidx() {
  State& s = st();
  return s.fibers[s.cur].tid;
}

}  // namespace simt

#define threadIdx (simt::tidx())
#define blockIdx (simt::st().bid)
#define blockDim (simt::st().bdim)
#define gridDim (simt::st().gdim)
#define __syncthreads() simt::syncthreads()
#define __shfl_xor(v, mask, ...) simt::shfl_idx((v), (simt::st().cur % simt::kWave) ^ (mask))
#define __shfl(v, lane, ...) simt::shfl_idx((v), (lane))
#define __shfl_down(v, d, ...) simt::shfl_idx((v), (simt::st().cur % simt::kWave) + (d))
#define __builtin_amdgcn_mfma_f32_32x32x2f32 simt::mfma_32x32x2f32
#define __builtin_amdgcn_mfma_f32_16x16x4f32 simt::mfma_16x16x4f32
#define __builtin_amdgcn_mfma_f64_16x16x4f64 simt::mfma_16x16x4f64
// Hardware approximations (v_rcp_f32, v_sqrt_f32, v_exp_f32: ~1 ulp on gfx950).  The emulator evaluates them exactly; setting
// UGLAD_EMUL_ULP_NOISE (bit 0: rcp, bit 1: sqrt, bit 2: exp2) in the environment moves the result one ulp up or down, by a
// hash of its bits -- a way to find out on the CPU which approximation a result is sensitive to.
namespace simt {
inline int ulp_noise_mask() {
  static const int m = [] {
    const char* e = std::getenv("UGLAD_EMUL_ULP_NOISE");
    return e ? std::atoi(e) : 0;
  }();
  return m;
}
inline float ulp_noise(float v, int bit) {
  if (!(ulp_noise_mask() & bit) || !(v == v) || v == 0.f || std::isinf(v)) return v;
  unsigned u;
  std::memcpy(&u, &v, 4);
  unsigned h = u * 2654435761u;
  h ^= h >> 15;
  const unsigned r = (h >> 7) % 3;  // 0: as is, 1: one ulp up, 2: one ulp down (in magnitude)
  if (r == 1) u += 1;
  if (r == 2) u -= 1;
  std::memcpy(&v, &u, 4);
  return v;
}
}  // namespace simt
#define __builtin_amdgcn_rcpf(x) (simt::ulp_noise(1.0f / (x), 1))
#define __builtin_amdgcn_sqrtf(x) (simt::ulp_noise(sqrtf(x), 2))
#define __builtin_amdgcn_rsqf(x) (simt::ulp_noise(1.0f / sqrtf(x), 2))
#define __builtin_amdgcn_logf(x) (simt::ulp_noise(log2f(x), 4))  /* v_log_f32 = log2 */
#define __builtin_amdgcn_readfirstlane(v) (v)  /* only ever applied to wave-uniform values */
#define __builtin_amdgcn_exp2f(x) (simt::ulp_noise(exp2f(x), 4))
#define __expf(x) expf(x)
#define __builtin_amdgcn_update_dpp simt::update_dpp
#define __builtin_amdgcn_readlane(v, l) simt::shfl_idx((v), (l))
#define __builtin_amdgcn_s_memtime() 0ull
#define __builtin_amdgcn_sched_barrier(m) ((void)0)
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __builtin_amdgcn_s_setprio(n) ((void)0) /* issue priority: no effect on results */
inline int atomicMax(int* p, int v) { const int o = *p; if (v > o) *p = v; return o; }
inline int atomicOr(int* p, int v) { const int o = *p; *p = o | v; return o; }
inline int atomicAdd(int* p, int v) { const int o = *p; *p = o + v; return o; }
inline float atomicAdd(float* p, float v) { const float o = *p; *p = o + v; return o; }
#define HIP_SYMBOL(x) (&(x))
inline hipError_t hipMemcpyFromSymbol(void* dst, const void* sym, size_t n) { memcpy(dst, sym, n); return hipSuccess; }


template <class K, class... Args>
inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t, hipStream_t, Args... args) {
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned b = 0; b < grid.x; ++b) simt::run_block(grid, block, dim3(b, by, bz), [&]() { kernel(args...); });
}
