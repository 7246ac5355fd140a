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
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

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

namespace simt {

constexpr int kWave = 64;
constexpr size_t kStack = 256 * 1024;

struct Fiber {
  ucontext_t ctx;
  char* stack = nullptr;
  bool done = false;
  const volatile unsigned* wait_ptr = nullptr;
  unsigned wait_val = 0;
  unsigned n_coll = 0;  // wave-collectives executed (parity selects the exchange buffer)
  dim3 tid;
};

struct WaveState {
  uint32_t slot[2][kWave];
  uint32_t slot2[2][kWave];
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

template <class T>
inline T shfl_idx(T v, int src_lane) {
  static_assert(sizeof(T) == 4, "32-bit shuffles only");
  State& s = st();
  WaveState& w = my_wave();
  Fiber& f = s.fibers[s.cur];
  const int lane = s.cur % kWave, par = f.n_coll++ & 1;
  memcpy(&w.slot[par][lane], &v, 4);
  wave_sync();
  T r;
  memcpy(&r, &w.slot[par][src_lane & (kWave - 1)], 4);
  return r;
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

inline void fiber_entry() {
  State& s = st();
  s.body();
  s.fibers[s.cur].done = true;
  ++s.progress;
  swapcontext(&s.fibers[s.cur].ctx, &s.sched);
}

inline void run_block(dim3 grid, dim3 block, unsigned bx, const std::function<void()>& body) {
  State& s = st();
  const int n = (int)(block.x * block.y * block.z);
  if ((int)s.fibers.size() < n) {
    const size_t old = s.fibers.size();
    s.fibers.resize(n);
    for (size_t i = old; i < (size_t)n; ++i) s.fibers[i].stack = (char*)malloc(kStack);
  }
  s.nthreads = n;
  s.arrived = 0;
  s.bid = dim3(bx, 0, 0);
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
    f.tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = kStack;
    f.ctx.uc_link = &s.sched;
    makecontext(&f.ctx, (void (*)())fiber_entry, 0);
  }
  int remaining = n;
  while (remaining > 0) {
    const unsigned long before = s.progress;
    bool ran = false;
    for (int i = 0; i < n; ++i) {
      Fiber& f = s.fibers[i];
      if (f.done) continue;
      if (f.wait_ptr && *f.wait_ptr == f.wait_val) continue;
      f.wait_ptr = nullptr;
      s.cur = i;
      ran = true;
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
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))

template <class K, class... Args>
inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t, hipStream_t, Args... args) {
  for (unsigned b = 0; b < grid.x; ++b) simt::run_block(grid, block, b, [&]() { kernel(args...); });
}
