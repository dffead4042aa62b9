// Shared device-side definitions of the fused MaCowUnit kernels (mcf_unit.hip: one workgroup per sample; mcf_unit_split.hip:
// a sample's 8x8 latent split by rows over 2 or 4 workgroups that exchange halo rows inside the launch).
#pragma once
#include <cstring>
#include <type_traits>

#include "mcf_dev.h"

namespace ipoke {

struct UnitLayer {
  const void* W1; const void* W2; const float* bias2;       // forward operands
  const void* W1T; const void* W2T;                         // backward operands
  float* y;                                                 // fwd: output state of this layer (NULL: not stored)
  void* a2_save; float* scale_save; float* ld_slot;
  const float* post_ls; const float* post_bias;             // ActNorm behind this layer (NULL: none)
  const float* x;                                           // bwd / inverse: saved input state of this layer
  const float* y_post; float* post_part;                    // bwd of the ActNorm: its saved output, [B][2C] partial sums
  void* dparams_save; void* dc_save; float* dbias_part;
  void* x_op;                                               // bwd: dtype [B*64][Cp] copy of x for the shifted-conv weight gradient (NULL: none)
  void* zc; int zc_off, zc_stride, zc_cin, zc_ld;           // fwd: conditioning operand of the coupling behind the unit (NULL: none)
  int order;
};
// The ActNorm2dFlow (+ Shuffle) and the NICE coupling in FRONT of the unit (forward order: coupling -> ActNorm -> unit, MaCowStep /
// macow2.py:1066-1117), differentiated by the unit's row-split backward launch on the rows each workgroup already holds
// (ipoke_unit_pair_desc; the work of ipoke_actnorm_affine_bwd)
struct UnitPair {
  const float* an_ls; const int* an_idx; const float* an_x; float* an_part;
  const float* x0; const float* scale; void* dparams; float* dbias_part; float* dx;
  int Cp, t_off, t_stride, ldp, on;
};
struct UnitParams {
  UnitLayer L[4];
  const float* x; const void* cond; const float* dy; const float* dld; float* dx;
  int ld, C, B, Cc, H, Cp, K1p, K2p, K3p, Hq, slot_w;
  unsigned long long* xchg; int xchg_stride;       // row-split launches (mcf_unit_split.hip): granule scratch, granules per (sample, layer, part)
  UnitPair pair;                                   // row-split backward only
#ifdef IPOKE_UNIT_STAMPS
  unsigned long long* stamps;       // probe build only (scripts/exp): shader-clock stamps of block 0
#endif
};
#ifdef IPOKE_UNIT_STAMPS
#define UNIT_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && U.stamps && (i) < 64) U.stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
// row-split kernels: the first four workgroups stamp, 64 slots each
#define UNIT_STAMP_S(i) do { if (blockIdx.x < 4 && threadIdx.x == 0 && U.stamps && (i) < 64) U.stamps[blockIdx.x * 64 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define UNIT_STAMP(i) do { } while (0)
#define UNIT_STAMP_S(i) do { } while (0)
#endif

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// The unit kernels take ~800 bytes of arguments (four layers of pointers), which hipcc fetches lazily: an s_load next to each first
// use, each behind an s_waitcnt.  The argument block of a launch is cold (the command processor has just written it), so that every
// first touch of a 64-byte line is a memory round trip of ~0.3 us and the prologue was a CHAIN of a dozen of them (stamps: 8 800
// cycles from entry to the staged tiles, no vector load waiting).  One dword of every line requested back to back at entry: one
// round trip, after which the compiler's own loads hit the scalar cache.
__device__ __forceinline__ void unit_kernarg_prefetch() {
  const auto ka = __builtin_amdgcn_kernarg_segment_ptr();
  unsigned d0, d1, d2, d3, d4, d5, d6, d7, d8, d9, d10, d11, d12;
  asm volatile(
      "s_load_dword %0, %13, 0x0\n\ts_load_dword %1, %13, 0x40\n\ts_load_dword %2, %13, 0x80\n\ts_load_dword %3, %13, 0xc0\n\t"
      "s_load_dword %4, %13, 0x100\n\ts_load_dword %5, %13, 0x140\n\ts_load_dword %6, %13, 0x180\n\ts_load_dword %7, %13, 0x1c0\n\t"
      "s_load_dword %8, %13, 0x200\n\ts_load_dword %9, %13, 0x240\n\ts_load_dword %10, %13, 0x280\n\ts_load_dword %11, %13, 0x2c0\n\t"
      "s_load_dword %12, %13, 0x300\n\ts_waitcnt lgkmcnt(0)"
      : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4), "=&s"(d5), "=&s"(d6), "=&s"(d7), "=&s"(d8), "=&s"(d9), "=&s"(d10), "=&s"(d11),
        "=&s"(d12)
      : "s"(ka)
      : "memory");
}
static_assert(sizeof(UnitParams) >= 0x304, "unit_kernarg_prefetch reads one dword of every 64-byte line up to 0x300");

// Weight fragments are fetched through buffer descriptors: the base lives in SGPRs, every lane needs ONE 32-bit byte offset
// per fragment column (instead of a 64-bit address per fragment, which the compiler kept live across the layer loop and
// spilled), the tap / K-step part of the address is a scalar offset, and rows beyond the matrix read as zero (hardware
// range check) so that no load sits behind a branch.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
static constexpr int kOob = 0x40000000;         // scalar offset beyond any weight matrix: the load returns zeros
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
template <typename T>
__device__ __forceinline__ typename ET<T>::frag buf_frag(rsrc_t rs, int voff, int soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
  return __builtin_bit_cast(typename ET<T>::frag, v);
}

// Width classes: every loop bound of the matrix-core phases is a compile-time constant of the class, rows / K steps a
// narrower layer does not have read as zero weights (range-checked loads) against zero-padded LDS tiles.  No branch sits
// inside a contraction, so the compiler pipelines LDS reads under the matrix cores.  (The per-layer kernels guard every
// fragment with `wave + 8 j < NF`, a per-lane condition to the compiler: each pair of MFMAs ends up in its own basic block.)
//   WIDE  : 32 < C <= 64  (Cp = 64, H <= 256, K2p <= 384, K3p <= 128, Hq <= 256)
//   narrow:      C <= 32  (Cp = 32, H <= 128, K2p <= 256, K3p <=  64, Hq <= 128)
template <bool WIDE> struct UC {
  static constexpr int CS = WIDE ? 2 : 1;       // 32-deep K steps per tap of the shifted conv
  static constexpr int J1 = WIDE ? 2 : 1;       // hidden-channel fragments per wave (8 waves x J1 x 16 >= H)
  static constexpr int N2S = WIDE ? 12 : 8;     // K steps of the 1x1 conv
  static constexpr int N3S = WIDE ? 4 : 2;      // K steps of its transpose (2C)
  static constexpr int HS = WIDE ? 8 : 4;       // K steps per tap of the shifted conv's transpose (4C)
};

// Operands are stored fragment-tiled (prep.hip: tiled_offset): fragment (row block rb, K step ks) of a matrix with nks steps
// per row is the KB at (rb * nks + ks) KB, lane l at byte 16 l.
template <typename T, bool WIDE>
__device__ __forceinline__ void unit_load_w1(McfW<T>& w, const void* W1, const UnitParams& U) {
  constexpr int KS = K64<T>::value;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nks = U.K1p / KS;
  const rsrc_t rs = make_rsrc(W1, ((U.H + 15) & ~15) * U.K1p * (int)sizeof(T));
  int voff[UC<WIDE>::J1];
#pragma unroll
  for (int j = 0; j < UC<WIDE>::J1; ++j) voff[j] = (wave + kMcfWaves * j) * nks * 1024 + lane * 16;
#pragma unroll
  for (int tap = 0; tap < 6; ++tap)
#pragma unroll
    for (int st = 0; st < UC<WIDE>::CS; ++st) {
      const int soff = (tap * UC<WIDE>::CS + st) * 1024;
#pragma unroll
      for (int j = 0; j < UC<WIDE>::J1; ++j) w.w1[tap][st][j] = buf_frag<T>(rs, voff[j], soff);
    }
}
template <typename T, bool WIDE>
__device__ __forceinline__ void unit_load_w2(McfW<T>& w, const void* W2, const UnitParams& U) {
  constexpr int KS = K64<T>::value;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n2 = U.K2p / KS;
  const rsrc_t rs = make_rsrc(W2, ((2 * U.C + 15) & ~15) * U.K2p * (int)sizeof(T));
  const int voff = wave * n2 * 1024 + lane * 16;
#pragma unroll
  for (int st = 0; st < UC<WIDE>::N2S; ++st)
    w.w2[st][0] = buf_frag<T>(rs, voff, st < n2 ? st * 1024 : kOob);   // unused K steps: out of range, zeros, no traffic
}

// LDS row pitches.  A fragment-shaped 16-byte access puts lane l = 16 gq + r on row r (+ a tap shift), 16-byte unit gq of a K step,
// and ds_read_b128 serves a wave in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS): a
// group holds eight rows at unit gq and the OTHER eight rows at unit gq + 1.  With a pitch of P 16-byte units the group's lanes fall on
// units P r + gq (mod 16): for an ODD P (the "+16 bytes" padding of rounds 1-2, chosen for sixteen consecutive lanes of one gq) seven
// of the eight rows at gq + 1 land on a unit that a row at gq already occupies -- every fragment read took 8 LDS cycles instead of 4
// (the unexplained 307 k / 309 k SQ_LDS_BANK_CONFLICT cycles per launch of round 2).  P = 2 (mod 4) is conflict free for every row
// shift (rows at gq take the even units, rows at gq + 1 the odd ones); all tile widths here are multiples of 64 bytes, so the pad is 32.
static constexpr int kTilePad = 32;
// Row pitch (floats) of the backward kernel's fp32 gradient tile: C rounded up to 4 floats, padded to 2 (mod 4) 16-byte units.
__host__ __device__ inline int unit_gb_pitch(int C) {
  const int u = (C + 3) >> 2;
  return (u + ((2 - u) & 3)) * 4;
}

// fp32 transforms of the bf16-net mode: hardware exp / log / rcp (1-2 ulp) instead of the libm-grade tanhf / logf / expm1f
// of the per-layer (parity-mode) kernels.  tanh(s/2) + 1 == 2 / (1 + exp(-s)); the ELU output is rounded to bf16 anyway.
__device__ __forceinline__ float fast_elu(float x) { return x > 0.f ? x : __expf(x) - 1.f; }
__device__ __forceinline__ float fast_scale(float s) { return __fdividef(2.f, 1.f + __expf(-s)); }

// row-split launches (mcf_unit_split.hip); S = 2 or 4 workgroups per sample.  Exchange scratch: 256 bytes of header, then per
// (sample, layer, consumer part) kUnitXchgStride 8-byte granules = [4 halo rows][8 columns][128 value pairs]
static constexpr int kUnitXchgStride = 4 * 8 * 128;
int unit_fwd_split_launch(const UnitParams& U, int S, hipStream_t s);
int unit_bwd_split_launch(const UnitParams& U, int S, hipStream_t s);

}  // namespace ipoke
