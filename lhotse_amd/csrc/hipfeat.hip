// libhipfeat: C ABI (include/hipfeat.h) + kernel dispatch.  gfx950 only; no torch, no Python.
#include "../../include/hipfeat.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "common.hpp"
#include "kernel_generic.hpp"
#include "fft512_common.hpp"
#include "kernel_fft512b.hpp"
#include "kernel_fft512c.hpp"
#include "kernel_fft1024c.hpp"
#include "kernel_fft2048c.hpp"
#include "mel4_schedule.hpp"
#include "kernel_resample.hpp"
#include "kernel_minibatch.hpp"
#include "kernel_specaug.hpp"
#include "kernel_whisper2.hpp"
#include "kernel_whisper3.hpp"
#include "kernel_fft256.hpp"
#include "kernel_fft256c.hpp"
#include "kernel_wave.hpp"
#include "host_bulk.hpp"

using namespace hipfeat;

// --------------------------------------------------------------------------------------
// error plumbing
// --------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

// Environment switches, two classes:
//   route_env  ROUTING switches choose among kernels / launch routes that ALL meet the parity bar (HIPFEAT_FORCE_GENERIC, *_VARIANT,
//              HIPFEAT_NO_FIXED_SCHEDULE, HIPFEAT_NO_WAVE_*, HIPFEAT_NO_FLAT, HIPFEAT_RESAMPLE_GENERIC, HIPFEAT_MB_NO_INLINE): the GPU
//              suite uses them to compare the instances of ONE library against each other, bit for bit where the arithmetic is the same.
//   exp_env    EXPERIMENT switches skip work or retune launch shapes (HIPFEAT_MB_SKIP = deliberately wrong results, HIPFEAT_MB_SLOTS,
//              HIPFEAT_ROUNDS, HIPFEAT_ROUNDS_R3).  They exist only in builds with -DHIPFEAT_EXPERIMENTS (tools/variants.py); in the
//              product library the names are never read, so no environment can make it emit wrong results.
static inline const char* route_env(const char* name) { return getenv(name); }
#ifdef HIPFEAT_EXPERIMENTS
static inline const char* exp_env(const char* name) { return getenv(name); }
#else
static inline const char* exp_env(const char*) { return nullptr; }
#endif

static hipfeat_status fail(hipfeat_status st, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return st;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return fail(HIPFEAT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorName(e_), __FILE__, __LINE__); \
  } while (0)

struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceGuard() {
    int cur = -1;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
  }
};

// --------------------------------------------------------------------------------------
// objects
// --------------------------------------------------------------------------------------
struct StagingSlot {
  void* h = nullptr;  // pinned host
  void* d = nullptr;  // device
  size_t cap = 0;
  hipEvent_t ev = nullptr;
  bool busy = false;
};

struct hipfeat_plan {
  hipfeat_config cfg{};
  int device = 0;
  int feature_dim = 0;
  int K = 0, H = 0, log2H = 0;
  bool pow2 = false;
  int npad_left = 0;
  std::string kernel_name = "generic";
  int blocks_per_cu = 0;  // hipOccupancyMaxActiveBlocksPerMultiprocessor for the selected kernel
  // device constants
  float* d_window = nullptr;
  float2* d_tw = nullptr;
  float* d_mel = nullptr;
  int2* d_mel_range = nullptr;
  float* d_dct = nullptr;
  float* d_lifter = nullptr;
  // generic kernel geometry
  int fpb = 8;
  int span = 0, off_z = 0, off_p = 0, off_tw = 0, off_stat = 0, off_mel = 0;
  size_t lds_bytes = 0;
  // fft512 fast path
  int variant = 0;  // 0 generic, 2 fft512 "b" (16-frame tiles), 4 fft256, 5 wave-per-frame, 6 whisper, 7 fft512 "c" (wave-autonomous fbank), 8 fft1024 "c", 9 whisper wave-autonomous + fused normalisation, 10 fft2048 "c", 11 fft256 "c"
  float* d_mel_a4 = nullptr;
  float* d_dct_consts = nullptr;
  bool fast_mfcc = false;
  int fast_out = 0;  // kernel b output stage: 0 fbank, 1 mfcc, 2 (log-)spectrogram
  int lm_stride = 0, dct_groups = 0, dct_floats = 0;
  int nrows = 0;    // template instance: pass-1 rows that can be non-zero
  int tiles_per_block = 4;
  int xs_floats = 0;
  size_t fast_lds_bytes = 0;
  int const_floats = 0;
  float* d_lds_consts = nullptr;
  float* d_mel_a = nullptr;
  WaveWork* d_work = nullptr;
  // fft512 wave-autonomous fbank kernel (variant 7)
  float* d_c_shared = nullptr;  // LDS image: FFT constants | 4x4-block filterbank weights | lane tables
  int c_shared_floats = 0, c_wtab_off = 0, c_ltab_off = 0, c_xs_floats = 0, c_rounds = 0, c_mode = 0;
  // wave-autonomous kernels: frames per workgroup = fpb_unit (frames of one round of all waves) x rounds, rounds chosen per LAYOUT between 2
  // and c_rounds_max: long launches take many rounds per workgroup (the constant tables and the first, un-overlapped span are paid once per
  // workgroup: 16 instead of 8 rounds is + 4 % on the bench workload), short ones few (enough workgroups to fill the chip)
  int fpb_unit = 0, c_rounds_max = 0;
  // fft1024 wave-autonomous fbank kernel (variant 8; shares d_c_shared / c_* with variant 7)
  int w_nsets = 0, w_steps[kWMaxSets] = {}, w_step0[kWMaxSets] = {};
  int w_waves = kWWaves;  // fft1024c: waves per workgroup (12 for the fixed-schedule instances)
  int w_fixed = 0;  // fft1024c / fft2048c: instance with a compile-time mel schedule (fft1024c_fixed_id)
  // fft2048 wave-autonomous fbank kernel (variant 10; shares d_c_shared / c_* / w_* with variants 7 and 8)
  float* d_x_twp = nullptr;  // [32][32] v2 W_1024^(q k1)
  int x_waves = 0, x_tws_off = 0, x_tw32_off = 0;
  bool x_odd = false;
  // wave-per-frame kernel (variant 5)
  float* d_mel_t = nullptr;  // filterbank blob (descriptors + compact weights)
  int mel_maxband = 0;
  int wave_blob_floats = 0;
  bool wave_dct_in_lds = false;
  size_t wave_lds_bytes = 0;
  // whisper FFT fast path (variant 6)
  float* d_wh2_cs = nullptr;
  float* d_wh2_tw = nullptr;
  float* d_wh2_mel = nullptr;
  int32_t* d_wh2_sched = nullptr;
  int32_t wh2_k0[kW2MaxMelTiles] = {}, wh2_steps[kW2MaxMelTiles] = {}, wh2_off[kW2MaxMelTiles] = {};
  int32_t wh2_wave_tiles[4][2] = {};
  // transient-layout staging ring (hipfeat_extract)
  mutable std::mutex mu;
  mutable StagingSlot slots[4];
  mutable int next_slot = 0;
  // host-form scratch (hipfeat_extract_host)
  mutable float* d_scratch_wave = nullptr;
  mutable size_t scratch_wave_cap = 0;
  mutable float* d_scratch_out = nullptr;
  mutable size_t scratch_out_cap = 0;
};

struct hipfeat_layout {
  int device = 0;
  int64_t batch = 0;
  int64_t total_frames = 0;
  int64_t total_blocks = 0;
  int64_t out_row_stride = 0;
  int uniform_bpc = 0;
  int fpb = 0, fpb_unit = 0;
  CutDesc* d_cuts = nullptr;
  bool owns = true;
  std::vector<int64_t> num_frames;
  std::vector<int32_t> block_cut;  // ragged batches: owner of every workgroup (uploaded behind the descriptors; uniform_bpc = -1)
  bool flat = false;               // ragged batch laid out by frame quads (fft512c FLAT instances): first_block = a cut's first quad
  int64_t total_quads = 0;
  // Whisper (variant 9): per-cut normalisation scratch behind the descriptors, kNormSlots copies handed out round-robin so that
  // launches of one layout that overlap on different streams do not share one (each copy re-arms itself at the end of its launch)
  // copy k: [total_blocks][2] float workgroup statistics, then [batch] uint32 completion counters (armed = 0)
  unsigned char* d_norm = nullptr;
  int norm_slots = 0;
  mutable std::atomic<unsigned> norm_next{0};
};
constexpr int kNormSlots = 4;
static size_t norm_slot_bytes(const hipfeat_layout* lay) { return (2 * (size_t)lay->total_blocks + (size_t)lay->batch) * 4; }

// --------------------------------------------------------------------------------------
// library / pure helpers
// --------------------------------------------------------------------------------------
extern "C" HIPFEAT_API int32_t hipfeat_abi_version(void) { return HIPFEAT_ABI_VERSION; }
extern "C" HIPFEAT_API const char* hipfeat_last_error(void) { return g_err; }

extern "C" HIPFEAT_API hipfeat_status hipfeat_device_count(int32_t* count) {
  if (!count) return fail(HIPFEAT_ERR_INVALID, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(HIPFEAT_ERR_HIP, "hipGetDeviceCount failed: %s", hipGetErrorName(e));
  }
  *count = n;
  return HIPFEAT_OK;
}

extern "C" HIPFEAT_API int64_t hipfeat_num_frames(int64_t num_samples, int32_t frame_length, int32_t frame_shift,
                                      int32_t snip_edges) {
  if (frame_shift <= 0 || frame_length <= 0 || num_samples < 0) return 0;
  if (snip_edges) return num_samples < frame_length ? 0 : 1 + (num_samples - frame_length) / frame_shift;
  return (num_samples + frame_shift / 2) / frame_shift;
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_check_length(int64_t padded_len, int32_t frame_length, int32_t frame_shift,
                                               int32_t snip_edges) {
  if (snip_edges) return HIPFEAT_OK;
  const int64_t t = hipfeat_num_frames(padded_len, frame_length, frame_shift, 0);
  if (t <= 0)
    return fail(HIPFEAT_ERR_TOO_SHORT, "waveform of %lld samples yields no frames", (long long)padded_len);
  const int64_t npad_left = (frame_length - frame_shift) / 2;
  const int64_t npad_right = (t - 1) * frame_shift + frame_length - padded_len - npad_left;
  if (npad_left > padded_len || npad_right > padded_len)
    return fail(HIPFEAT_ERR_TOO_SHORT,
                "waveform of %lld samples is shorter than the reflect padding (%lld left, %lld right)",
                (long long)padded_len, (long long)npad_left, (long long)npad_right);
  return HIPFEAT_OK;
}

// --------------------------------------------------------------------------------------
// plan
// --------------------------------------------------------------------------------------
template <typename T>
static hipfeat_status upload(T** dst, const T* src, size_t n) {
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(dst), std::max<size_t>(n, 1) * sizeof(T)));
  if (n) HIP_TRY(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
  return HIPFEAT_OK;
}

static void plan_free(hipfeat_plan* p) {
  if (!p) return;
  DeviceGuard g(p->device);
  (void)hipFree(p->d_window);
  (void)hipFree(p->d_tw);
  (void)hipFree(p->d_mel);
  (void)hipFree(p->d_mel_range);
  (void)hipFree(p->d_wh2_cs);
  (void)hipFree(p->d_x_twp);
  (void)hipFree(p->d_wh2_tw);
  (void)hipFree(p->d_wh2_mel);
  (void)hipFree(p->d_wh2_sched);
  (void)hipFree(p->d_dct);
  (void)hipFree(p->d_lifter);
  (void)hipFree(p->d_scratch_wave);
  (void)hipFree(p->d_scratch_out);
  (void)hipFree(p->d_lds_consts);
  (void)hipFree(p->d_mel_a4);
  (void)hipFree(p->d_dct_consts);
  (void)hipFree(p->d_mel_a);
  (void)hipFree(p->d_work);
  (void)hipFree(p->d_mel_t);
  (void)hipFree(p->d_c_shared);
  for (auto& s : p->slots) {
    if (s.h) (void)hipHostFree(s.h);
    if (s.d) (void)hipFree(s.d);
    if (s.ev) (void)hipEventDestroy(s.ev);
  }
  delete p;
}

// --------------------------------------------------------------------------------------
// fft512 fast path: eligibility, constants, mel work split
// --------------------------------------------------------------------------------------
template <int NROWS, int OUT>
static const void* fft512b_entry() {
  return reinterpret_cast<const void*>(&fft512b_kernel<NROWS, OUT>);
}

// The dynamic-LDS limit of a kernel is a property of the FUNCTION, shared by every plan of the process: it is only ever
// raised (per device), so that a later plan with a smaller footprint cannot lower it under an earlier plan's launches.
#ifdef HIPFEAT_LDS_POISON
static void set_lds_poison(size_t bytes) {
  const int n = (int)(bytes / sizeof(float));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(hipfeat::g_lds_poison_floats), &n, sizeof(n));
}
#else
static inline void set_lds_poison(size_t) {}
#endif

static hipError_t ensure_dynamic_lds(const void* fn, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, size_t> high;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(mu);
  size_t& cur = high[{fn, dev}];
  if (bytes <= cur) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) cur = bytes;
  return e;
}

// Mel work split of the fast kernels: band of every 16-mel tile in 8-bin groups, assigned to the 4 waves, and the MFMA
// A operands in lane order.  Returns false when the filterbank does not fit the static schedule (-> generic kernel).
static bool build_mel_schedule(const float* h_mel, int M, int K, int prow_stride, int ntiles, WaveWork (&work)[4], std::vector<float>& mel_a) {
  struct Seg { int tile, bin, ng; };
  std::vector<Seg> segs;
  for (int t = 0; t < ntiles; ++t) {
    int lo = K, hi = 0;
    for (int k = 0; k < K; ++k)
      for (int j = 16 * t; j < std::min(M, 16 * t + 16); ++j)
        if (h_mel[(size_t)k * M + j] != 0.0f) {
          lo = std::min(lo, k);
          hi = std::max(hi, k + 1);
        }
    if (hi == 0) lo = 0, hi = 1;
    int lo2 = lo & ~1;
    int ng = (hi - lo2 + 7) / 8;
    if (lo2 + 8 * ng > prow_stride) lo2 = (prow_stride - 8 * ng) & ~1;  // keep reads inside the padded row
    if (lo2 < 0 || ng > kMaxGroups0) return false;
    segs.push_back({t, lo2, ng});
  }
  // the four widest tiles become the waves' first segment, the rest go to the least loaded waves
  std::sort(segs.begin(), segs.end(), [](const Seg& a, const Seg& b) { return a.ng > b.ng; });
  std::memset(work, 0, sizeof(work));
  int load[4] = {0, 0, 0, 0};
  bool has1[4] = {false, false, false, false};
  for (size_t i = 0; i < segs.size(); ++i) {
    const Seg& sg = segs[i];
    if (i < 4) {
      work[i].tile0 = sg.tile; work[i].bin0 = sg.bin; work[i].ngroups0 = sg.ng;
      load[i] = sg.ng;
      continue;
    }
    if (sg.ng > kMaxGroups1) return false;
    int best = -1;
    for (int w = 0; w < 4; ++w)
      if (!has1[w] && (best < 0 || load[w] < load[best])) best = w;
    if (best < 0) return false;
    work[best].tile1 = sg.tile; work[best].bin1 = sg.bin; work[best].ngroups1 = sg.ng;
    has1[best] = true;
    load[best] += sg.ng;
  }
  // MFMA A operands: lane (i = lane & 15, kk = lane >> 4) of step (2*gi + r) holds
  // W[bin + 8*gi + 2*kk + r][16*tile + i]; the second segment's steps start at 2*kMaxGroups0
  mel_a.assign((size_t)4 * kMelARegs * 64, 0.0f);
  for (int w = 0; w < 4; ++w)
    for (int sgm = 0; sgm < 2; ++sgm) {
      const int tile = sgm ? work[w].tile1 : work[w].tile0, bin = sgm ? work[w].bin1 : work[w].bin0;
      const int ng = sgm ? work[w].ngroups1 : work[w].ngroups0;
      for (int g2 = 0; g2 < ng; ++g2)
        for (int r = 0; r < 2; ++r)
          for (int lane = 0; lane < 64; ++lane) {
            const int i = lane & 15, kk = lane >> 4;
            const int b = bin + 8 * g2 + 2 * kk + r, m = 16 * tile + i;
            const int step = 2 * ((sgm ? kMaxGroups0 : 0) + g2) + r;
            if (b < K && m < M) mel_a[((size_t)w * kMelARegs + step) * 64 + lane] = h_mel[(size_t)b * M + m];
          }
    }
  return true;
}

// DCT^T as MFMA A operands + lifter (MFCC stage of the fast kernels): lane (i = lane & 15, kk = lane >> 4) of group g,
// half r holds dct[mel = 8 g + 2 kk + r][ceps = 16 ct + i]  (Wav2MFCC._dct, layers.py:697-706)
static std::vector<float> build_dct_operands(const hipfeat_config& c, const float* h_dct, const float* h_lifter, int dct_groups) {
  const int M = c.num_filters, C = c.num_ceps, nct = (C + 15) / 16;
  std::vector<float> da((size_t)nct * dct_groups * 64 * 2, 0.0f);
  for (int ct = 0; ct < nct; ++ct)
    for (int g2 = 0; g2 < dct_groups; ++g2)
      for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 2; ++r) {
          const int i = lane & 15, kk = lane >> 4, m = 8 * g2 + 2 * kk + r, cc = 16 * ct + i;
          if (m < M && cc < C) da[(((size_t)ct * dct_groups + g2) * 64 + lane) * 2 + r] = h_dct[(size_t)m * C + cc];
        }
  for (int cc = 0; cc < 64; ++cc)  // lifter (layers.py:681-695), ones when cepstral_lifter == 0
    da.push_back((c.apply_lifter && h_lifter && cc < C) ? h_lifter[cc] : 1.0f);
  return da;
}

// --------------------------------------------------------------------------------------
// fft512 wave-autonomous fbank kernel (kernel_fft512c.hpp); its filterbank schedule is built in mel4_schedule.hpp
// --------------------------------------------------------------------------------------
template <int NROWS, int NFULL, int MODE, bool FLAT = false>
static const void* fft512c_entry() {
  return reinterpret_cast<const void*>(&fft512c_kernel<NROWS, NFULL, MODE, FLAT>);
}

// (rows, frame length, mode) -> kernel instance; `launch` == false only returns the entry point
// the (rows, frame length) that have a FLAT instance (ragged batches by frame quads, kernel_fft512c.hpp): the 25 ms @ 16 kHz default
static bool fft512c_has_flat(int nrows, int N) { return nrows == 13 && N >= 384; }

template <int MODE>
static const void* fft512c_pick(int nrows, int N, bool launch, dim3 grid, dim3 block, size_t lds, hipStream_t stream, const Fft512cParams* fp, bool flat = false) {
  // 25 ms at 16 kHz (N = 400: 12 full rows of 32 samples + a partial one) gets the instance without length masks on the full rows
#define HF_C_CASE(R, F)                                                                                   \
  {                                                                                                       \
    if (launch) hipLaunchKernelGGL((fft512c_kernel<R, F, MODE>), grid, block, lds, stream, *fp);          \
    return fft512c_entry<R, F, MODE>();                                                                   \
  }
  if (flat && fft512c_has_flat(nrows, N)) {
    if (launch) hipLaunchKernelGGL((fft512c_kernel<13, 12, MODE, true>), grid, block, lds, stream, *fp);
    return fft512c_entry<13, 12, MODE, true>();
  }
  if (nrows == 10) HF_C_CASE(10, 0)
  if (nrows == 13 && N >= 384) HF_C_CASE(13, 12)
  if (nrows == 13) HF_C_CASE(13, 0)
  HF_C_CASE(16, 0)
#undef HF_C_CASE
}
static const void* fft512c_dispatch(int mode, int nrows, int N, bool launch, dim3 grid, dim3 block, size_t lds, hipStream_t stream, const Fft512cParams* fp,
                                    bool flat = false) {
  return mode == 0 ? fft512c_pick<0>(nrows, N, launch, grid, block, lds, stream, fp, flat)
                   : (mode == 1 ? fft512c_pick<1>(nrows, N, launch, grid, block, lds, stream, fp, flat)
                                : (mode == 2 ? fft512c_pick<2>(nrows, N, launch, grid, block, lds, stream, fp, flat)
                                             : fft512c_pick<3>(nrows, N, launch, grid, block, lds, stream, fp, flat)));
}

// Returns HIPFEAT_OK with p->variant == 7 when the configuration takes the wave-autonomous kernel, HIPFEAT_OK with the
// variant untouched when it does not (the caller then sets up kernel "b").
static hipfeat_status setup_fft512c(hipfeat_plan* p, const float* h_window, const float* h_mel, int nrows, const float* h_dct, const float* h_lifter) {
  const hipfeat_config& c = p->cfg;
  const int N = c.frame_length, shift = c.frame_shift, M = c.num_filters;
  const bool mfcc = c.kind == HIPFEAT_MFCC;
  if (mfcc && (M > 4 * kCDctChunks || c.num_ceps > 64 || !h_dct)) return HIPFEAT_OK;
  // mode 0: 2 accumulator sets x 16 steps (many narrow filters); modes 1 / 2 (MFCC): 1 set x 32 steps (few, wide filters)
  Mel4Schedule sch;
  int mode = mfcc ? (M <= 4 * kCDctChunksSmall ? 3 : 2) : 0;  // MFCC: 3 = at most 24 filters (6 chunks of DCT operands + split-step twiddles in registers)
  if (mfcc || !build_mel4_schedule(h_mel, M, p->K, kCPRowStride, kCMaxSets, kCMaxSteps, sch)) {
    if (!build_mel4_schedule(h_mel, M, p->K, kCPRowStride, 1, 2 * kCMaxSteps, sch)) return HIPFEAT_OK;
    if (!mfcc) mode = 1;
  }
  const int tsets = mode == 0 ? kCMaxSets : 1, tsteps = mode == 0 ? kCMaxSteps : 2 * kCMaxSteps;
  // LDS image: window/2 as (even, odd) sample pairs per (row n1, lane q); pass twiddles W_256^(q k1) per (row k1, lane q);
  // split-step twiddles -i W_512^(q + 16 k2) per (row k2 < 8, lane q); then the filterbank tables
  std::vector<float> img((size_t)(nrows * 16 + 256 + 128) * 2, 0.0f);
  for (int n1 = 0; n1 < nrows; ++n1)
    for (int q = 0; q < 16; ++q)
      for (int e = 0; e < 2; ++e) {
        const int i = 32 * n1 + 2 * q + e;
        img[2 * (n1 * 16 + q) + e] = i < N ? 0.5f * h_window[i] : 0.0f;
      }
  float* twp = img.data() + 2 * nrows * 16;
  float* tws = twp + 512;
  for (int k1 = 0; k1 < 16; ++k1)
    for (int q = 0; q < 16; ++q) {
      const double a = -2.0 * M_PI * (double)(q * k1) / 256.0;
      twp[2 * (k1 * 16 + q)] = (float)std::cos(a);
      twp[2 * (k1 * 16 + q) + 1] = (float)std::sin(a);
    }
  for (int k2 = 0; k2 < 8; ++k2)
    for (int q = 0; q < 16; ++q) {  // w = -i * W_512^k = (sin(a), -cos(a)) with a = -2 pi k / 512
      const double a = -2.0 * M_PI * (double)(q + 16 * k2) / 512.0;
      tws[2 * (k2 * 16 + q)] = (float)std::sin(a);
      tws[2 * (k2 * 16 + q) + 1] = (float)(-std::cos(a));
    }
  // the kernel runs `tsets` sets of `tsteps` steps unconditionally: pad the tables (weights 0, no output column)
  p->c_wtab_off = (int)img.size();
  img.resize(img.size() + (size_t)tsets * tsteps * 64, 0.0f);
  for (int s2 = 0; s2 < sch.nsets; ++s2)
    std::memcpy(img.data() + p->c_wtab_off + (size_t)s2 * tsteps * 64, sch.wtab.data() + (size_t)sch.step0[s2] * 64, (size_t)sch.steps[s2] * 64 * sizeof(float));
  p->c_ltab_off = (int)img.size();
  img.resize(img.size() + (size_t)tsets * 256, 0.0f);
  {
    const int none = kMel4NoColumn;
    for (int s2 = 0; s2 < tsets; ++s2)
      for (int lane = 0; lane < 64; ++lane) {
        float* lt = img.data() + p->c_ltab_off + ((size_t)s2 * 64 + lane) * 4;
        if (s2 < sch.nsets) std::memcpy(lt, sch.ltab.data() + ((size_t)s2 * 64 + lane) * 4, 4 * sizeof(float));
        else std::memcpy(lt + 1, &none, 4);
      }
  }
  while (img.size() % 64) img.push_back(0.0f);
  p->c_shared_floats = (int)img.size();
  p->c_xs_floats = (3 * shift + 32 * nrows + 3) & ~3;
  const size_t lds = ((size_t)p->c_shared_floats + (size_t)kCWaves * (p->c_xs_floats + kCRegion)) * sizeof(float);
  if (lds > 80 * 1024 || (p->c_xs_floats >> 8) > 6) return HIPFEAT_OK;  // two workgroups of 8 waves per CU or nothing
  const void* fn = fft512c_dispatch(mode, nrows, N, false, dim3(), dim3(), 0, nullptr, nullptr);
  hipError_t e = ensure_dynamic_lds(fn, lds);
  if (e != hipSuccess) return fail(HIPFEAT_ERR_HIP, "hipFuncSetAttribute(fft512c) failed: %s", hipGetErrorName(e));
  hipfeat_status st;
  if ((st = upload(&p->d_c_shared, img.data(), img.size())) != HIPFEAT_OK) return st;
  if (mfcc) {  // DCT operands in matrix-core lane order: [chunk of 4 filters][lane = cepstral coefficient][filter in the chunk], then the lifter
    const int C = c.num_ceps;
    const int dch = mode == 3 ? kCDctChunksSmall : kCDctChunks;
    std::vector<float> dt((size_t)dch * 256 + 64, 0.0f);
    for (int m = 0; m < M; ++m)
      for (int cc = 0; cc < C; ++cc) dt[((size_t)(m / 4) * 64 + cc) * 4 + (m & 3)] = h_dct[(size_t)m * C + cc];
    for (int cc = 0; cc < 64; ++cc) dt[(size_t)dch * 256 + cc] = (c.apply_lifter && h_lifter && cc < C) ? h_lifter[cc] : 1.0f;
    if ((st = upload(&p->d_dct_consts, dt.data(), dt.size())) != HIPFEAT_OK) return st;
  }
  p->c_mode = mode;
  p->nrows = nrows;
  p->c_rounds = 8;  // 8 waves x 8 rounds x 4 frames = 256 frames per workgroup
  p->fpb = kCWaves * p->c_rounds * 4;
  p->fpb_unit = kCWaves * 4;
  p->c_rounds_max = 16;
  p->fast_lds_bytes = lds;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64 * kCWaves, lds) == hipSuccess) p->blocks_per_cu = nb;
  int total_steps = 0;
  for (int s2 = 0; s2 < sch.nsets; ++s2) total_steps += sch.steps[s2];
  char nm[128];
  snprintf(nm, sizeof(nm), "fft512c_kernel<%d> %s lds=%zuB blocks/CU=%d mel4=%dx%d", nrows, mfcc ? "mfcc" : "fbank", lds, p->blocks_per_cu, sch.nsets, total_steps);
  p->kernel_name = nm;
  p->variant = 7;
  return HIPFEAT_OK;
}

static hipfeat_status setup_fft512(hipfeat_plan* p, const float* h_window, const float* h_mel, const float* h_dct,
                                   const float* h_lifter) {
  const hipfeat_config& c = p->cfg;
  const char* force = route_env("HIPFEAT_FORCE_GENERIC");
  if (force && force[0] == '1') return HIPFEAT_OK;
  const int N = c.frame_length, shift = c.frame_shift, M = c.num_filters;
  const bool mfcc = c.kind == HIPFEAT_MFCC;
  const bool spec = c.kind == HIPFEAT_SPECTROGRAM || c.kind == HIPFEAT_LOG_SPECTROGRAM;
  if (c.kind > HIPFEAT_MFCC || c.fft_length != 512 || (shift & 1) || N < 32 || c.use_energy || (!spec && c.use_fft_mag))
    return HIPFEAT_OK;
  if (mfcc && (M > 8 * kMaxDctGroups || c.num_ceps > 64)) return HIPFEAT_OK;
  const int need = (N + 31) / 32;
  const int nrows = need <= 10 ? 10 : (need <= 13 ? 13 : 16);
  const int ntiles = spec ? 0 : (M + 15) / 16;
  if (ntiles > 8) return HIPFEAT_OK;

  if (!spec) {  // log-mel filterbank / MFCC: the wave-autonomous kernel, unless the schedule or the LDS budget says no
    const char* var = route_env("HIPFEAT_FFT512_VARIANT");
    if (!(var && var[0] == 'b')) {  // HIPFEAT_FFT512_VARIANT=b: the 16-frame-tile kernel for these too (tests compare the two)
      hipfeat_status stc = setup_fft512c(p, h_window, h_mel, nrows, h_dct, h_lifter);
      if (stc != HIPFEAT_OK || p->variant == 7) return stc;
    }
  }
  WaveWork work[4];
  std::vector<float> mel_a;
  if (!build_mel_schedule(h_mel, M, p->K, kPRowStride, ntiles, work, mel_a)) return HIPFEAT_OK;  // -> generic kernel
  // LDS constant block: window/2 as (even, odd) sample pairs per (row n1, lane q); pass twiddles
  // W_256^(q k1) per (row k1, lane q); split-step twiddles -i W_512^(q + 16 k2) per (row k2, lane q)
  std::vector<float> wh(512, 0.0f);
  for (int i = 0; i < N; ++i) wh[i] = 0.5f * h_window[i];
  std::vector<float> lc((size_t)(nrows * 16 + 256 + 128) * 2, 0.0f);
  for (int n1 = 0; n1 < nrows; ++n1)
    for (int q = 0; q < 16; ++q) {
      lc[2 * (n1 * 16 + q)] = wh[32 * n1 + 2 * q];
      lc[2 * (n1 * 16 + q) + 1] = wh[32 * n1 + 2 * q + 1];
    }
  float* twp = lc.data() + 2 * nrows * 16;
  float* tws = twp + 512;
  for (int k1 = 0; k1 < 16; ++k1)
    for (int q = 0; q < 16; ++q) {
      const double a = -2.0 * M_PI * (double)(q * k1) / 256.0;
      twp[2 * (k1 * 16 + q)] = (float)std::cos(a);
      twp[2 * (k1 * 16 + q) + 1] = (float)std::sin(a);
    }
  for (int k2 = 0; k2 < 8; ++k2)
    for (int q = 0; q < 16; ++q) {  // w = -i * W_512^k = (sin(a), -cos(a)) with a = -2 pi k / 512
      const double a = -2.0 * M_PI * (double)(q + 16 * k2) / 512.0;
      tws[2 * (k2 * 16 + q)] = (float)std::sin(a);
      tws[2 * (k2 * 16 + q) + 1] = (float)(-std::cos(a));
    }
  hipfeat_status st;
  if ((st = upload(&p->d_lds_consts, lc.data(), lc.size())) != HIPFEAT_OK) return st;
  if ((st = upload(&p->d_mel_a, mel_a.data(), mel_a.size())) != HIPFEAT_OK) return st;
  if ((st = upload(&p->d_work, work, 4)) != HIPFEAT_OK) return st;
  p->nrows = nrows;
  // 16 tiles (256 frames) per workgroup: the constant-table load and the first, un-overlapped span
  // fetch are paid once per workgroup (measured on MI355X: 4 -> 1.98 M, 8 -> 2.11 M, 16 -> 2.16 M cuts/s)
  p->tiles_per_block = 16;
  const int const_floats = (int)lc.size();
  p->const_floats = const_floats;
  const void* fn;
  {
    // weights as 16-byte vectors: [wave][step / 4][lane][step % 4]
    std::vector<float> mel_a4(mel_a.size());
    for (int w = 0; w < 4; ++w)
      for (int st = 0; st < kMelARegs; ++st)
        for (int lane = 0; lane < 64; ++lane)
          mel_a4[(((size_t)w * kBMelVec + st / 4) * 64 + lane) * 4 + (st & 3)] = mel_a[((size_t)w * kMelARegs + st) * 64 + lane];
    if ((st = upload(&p->d_mel_a4, mel_a4.data(), mel_a4.size())) != HIPFEAT_OK) return st;
    p->xs_floats = (15 * shift + 32 * nrows + 255) & ~255;  // whole 1 KiB LDS-DMA chunks
    size_t lds_floats = (size_t)p->xs_floats + const_floats + 4 * kBWaveRegion;
    if (mfcc) {
      p->dct_groups = (M + 7) / 8;
      p->lm_stride = ntiles <= 2 ? 36 : (ntiles <= 4 ? 68 : 132);  // 4 mod 32: conflict-free 8-byte reads of the log-mel tile; 36 keeps MFCC-13 at 4 workgroups/CU
      std::vector<float> da = build_dct_operands(c, h_dct, h_lifter, p->dct_groups);
      p->dct_floats = (int)da.size();
      if ((st = upload(&p->d_dct_consts, da.data(), da.size())) != HIPFEAT_OK) return st;
      lds_floats += (size_t)kTileFrames * p->lm_stride + da.size();
      p->fast_mfcc = true;
    }
    p->fast_lds_bytes = lds_floats * sizeof(float);
    p->fast_out = mfcc ? 1 : (spec ? 2 : 0);
    if (mfcc)
      fn = nrows == 10 ? fft512b_entry<10, 1>() : (nrows == 13 ? fft512b_entry<13, 1>() : fft512b_entry<16, 1>());
    else if (spec)
      fn = nrows == 10 ? fft512b_entry<10, 2>() : (nrows == 13 ? fft512b_entry<13, 2>() : fft512b_entry<16, 2>());
    else
      fn = nrows == 10 ? fft512b_entry<10, 0>() : (nrows == 13 ? fft512b_entry<13, 0>() : fft512b_entry<16, 0>());
  }
  if (p->fast_lds_bytes > 160 * 1024) return HIPFEAT_OK;
  hipError_t e = ensure_dynamic_lds(fn, p->fast_lds_bytes);
  if (e != hipSuccess) return fail(HIPFEAT_ERR_HIP, "hipFuncSetAttribute(fft512) failed: %s", hipGetErrorName(e));
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 256, p->fast_lds_bytes) == hipSuccess) p->blocks_per_cu = nb;
  char nm[96];
  // same spelling as the device symbol rocprofv3 reports (modulo the space after the comma)
  snprintf(nm, sizeof(nm), "fft512b_kernel<%d,%d> %s lds=%zuB blocks/CU=%d", nrows, p->fast_out,
           mfcc ? "mfcc" : (spec ? "spectrogram" : "fbank"), p->fast_lds_bytes, p->blocks_per_cu);
  p->kernel_name = nm;
  p->variant = 2;
  p->fpb = kTileFrames * p->tiles_per_block;
  return HIPFEAT_OK;
}

// --------------------------------------------------------------------------------------
// fft1024 wave-autonomous fbank kernel (kernel_fft1024c.hpp): 22.05 / 24 / 32 kHz Kaldi log-mel
// --------------------------------------------------------------------------------------
template <int NROWS, int S0 = 0, int S1 = 0, int S2 = 0, bool PLAIN = false>
static const void* fft1024c_entry() {
  return reinterpret_cast<const void*>(&fft1024c_kernel<NROWS, S0, S1, S2, PLAIN>);
}
// instances with the mel schedule as compile-time constants (kernel_fft1024c.hpp): 1 = <20, 24,16,8> (24 kHz), 2 = <26, 24,16,8> (32 kHz),
// 3 = <20, 24,24,8> (22.05 kHz), 4 = <32, 16,16,8, PLAIN> (the librosa default: n_fft 1024 @ 22.05 kHz, 80 slaney filters, no DC removal,
// no pre-emphasis); 0 = generic
static int fft1024c_fixed_id(int nrows, int nsets, const int* steps, bool plain) {
  if (nsets != 3 || route_env("HIPFEAT_NO_FIXED_SCHEDULE")) return 0;
  if (nrows == 32) return plain && steps[0] == 16 && steps[1] == 16 && steps[2] == 8 ? 4 : 0;
  if (nrows == 20 && steps[0] == 24 && steps[1] == 16 && steps[2] == 8) return 1;
  if (nrows == 26 && steps[0] == 24 && steps[1] == 16 && steps[2] == 8) return 2;
  if (nrows == 20 && steps[0] == 24 && steps[1] == 24 && steps[2] == 8) return 3;
  return 0;
}

static hipfeat_status setup_fft1024c(hipfeat_plan* p, const float* h_window, const float* h_mel) {
  const hipfeat_config& c = p->cfg;
  const int N = c.frame_length, shift = c.frame_shift, M = c.num_filters;
  const bool librosa = c.kind == HIPFEAT_LIBROSA_FBANK;  // centred frames, |X| or |X|^2, log10 (librosa_fbank.py:66-137)
  if (p->variant != 0 || (c.kind != HIPFEAT_FBANK && !librosa) || c.fft_length != 1024 || (shift & 1) || N < 32 * 17 || c.use_energy ||
      (c.use_fft_mag && !librosa) || route_env("HIPFEAT_FORCE_GENERIC") || route_env("HIPFEAT_NO_WAVE_AUTONOMOUS"))
    return HIPFEAT_OK;
  const int need = (N + 31) / 32;
  const int nrows = need <= 20 ? 20 : (need <= 26 ? 26 : 32);
  Mel4Schedule sch;
  if (!build_mel4_schedule(h_mel, M, p->K, kWPRowStride, kWMaxSets, kWMaxSteps, sch)) return HIPFEAT_OK;
  std::vector<float> img((size_t)(nrows * 16 + 512 + kWSplitSteps * 16) * 2, 0.0f);
  for (int n1 = 0; n1 < nrows; ++n1)
    for (int q = 0; q < 16; ++q)
      for (int e = 0; e < 2; ++e) {
        const int i = 32 * n1 + 2 * q + e;
        img[2 * (n1 * 16 + q) + e] = i < N ? 0.5f * h_window[i] : 0.0f;
      }
  float* twp = img.data() + 2 * nrows * 16;
  float* tws = twp + 1024;
  for (int k1 = 0; k1 < 32; ++k1)
    for (int q = 0; q < 16; ++q) {
      const double a = -2.0 * M_PI * (double)(q * k1) / 512.0;
      twp[2 * (k1 * 16 + q)] = (float)std::cos(a);
      twp[2 * (k1 * 16 + q) + 1] = (float)std::sin(a);
    }
  for (int st = 0; st < kWSplitSteps; ++st)
    for (int q = 0; q < 16; ++q) {  // bin of the step's first operand: lanes >= 1: q + 32 s; lane 0: 32 s (s <= 8), 16 + 32 (s - 9) (s <= 15), 240
      int k;
      if (q != 0) k = st < 16 ? q + 32 * st : 0;
      else k = st <= 8 ? 32 * st : (st <= 15 ? 16 + 32 * (st - 9) : 240);
      const double a = -2.0 * M_PI * (double)k / 1024.0;  // w = -i * W_1024^k = (sin(a), -cos(a))
      tws[2 * (st * 16 + q)] = (float)std::sin(a);
      tws[2 * (st * 16 + q) + 1] = (float)(-std::cos(a));
    }
  // the kernel runs two accumulation chains per set over chunks of 4 steps: every set's steps are padded to a multiple of 8
  // (zero weights; the power-row reads stay inside the wave's region)
  p->c_wtab_off = (int)img.size();
  {
    int step0 = 0;
    for (int s2 = 0; s2 < sch.nsets; ++s2) {
      const int padded = (sch.steps[s2] + 7) & ~7;
      const size_t at = img.size();
      img.resize(at + (size_t)padded * 64, 0.0f);
      std::memcpy(img.data() + at, sch.wtab.data() + (size_t)sch.step0[s2] * 64, (size_t)sch.steps[s2] * 64 * sizeof(float));
      sch.steps[s2] = padded;
      sch.step0[s2] = step0;
      step0 += padded;
    }
  }
  p->c_ltab_off = (int)img.size();
  img.insert(img.end(), sch.ltab.begin(), sch.ltab.end());
  while (img.size() % 64) img.push_back(0.0f);
  p->c_shared_floats = (int)img.size();
  p->c_xs_floats = (3 * shift + 32 * nrows + 3) & ~3;
  // fixed-schedule instances: 12 waves per workgroup, the span buffer aliases the exchange / power region (kernel_fft1024c.hpp)
  int fixed = fft1024c_fixed_id(nrows, sch.nsets, sch.steps, !c.remove_dc_offset && c.preemph_coeff == 0.0f);
  if (fixed && p->c_xs_floats > kWRegion) fixed = 0;
  if (fixed >= 1 && fixed <= 3 && (librosa || c.use_fft_mag)) fixed = 0;  // the Kaldi instances have |X|^2, ln and Kaldi's edges compiled in
  // ... and the default frame geometry at 24 / 32 / 22.05 kHz with 80 filters (kernel_fft1024c.hpp)
  if (fixed >= 1 && fixed <= 3 && !(M == 80 && !c.snip_edges && ((fixed == 1 && N == 600 && shift == 240) || (fixed == 2 && N == 800 && shift == 320) || (fixed == 3 && N == 551 && shift == 220))))
    fixed = 0;
  const int waves = fixed ? kWWavesFixed : kWWaves;
  const size_t lds = ((size_t)p->c_shared_floats + (size_t)waves * (fixed ? kWRegion : p->c_xs_floats + kWRegion)) * sizeof(float);
  if (lds > 160 * 1024 || (p->c_xs_floats >> 8) > 10) return HIPFEAT_OK;
  const void* fn = fixed == 1 ? fft1024c_entry<20, 24, 16, 8>()
                   : fixed == 2 ? fft1024c_entry<26, 24, 16, 8>()
                   : fixed == 3 ? fft1024c_entry<20, 24, 24, 8>()
                   : fixed == 4 ? fft1024c_entry<32, 16, 16, 8, true>()
                   : nrows == 20 ? fft1024c_entry<20>() : (nrows == 26 ? fft1024c_entry<26>() : fft1024c_entry<32>());
  hipError_t e = ensure_dynamic_lds(fn, lds);
  if (e != hipSuccess) return fail(HIPFEAT_ERR_HIP, "hipFuncSetAttribute(fft1024c) failed: %s", hipGetErrorName(e));
  p->w_fixed = fixed;
  hipfeat_status st;
  if ((st = upload(&p->d_c_shared, img.data(), img.size())) != HIPFEAT_OK) return st;
  p->w_nsets = sch.nsets;
  int total_steps = 0;
  for (int s2 = 0; s2 < kWMaxSets; ++s2) {
    p->w_steps[s2] = s2 < sch.nsets ? sch.steps[s2] : 0;
    p->w_step0[s2] = s2 < sch.nsets ? sch.step0[s2] : 0;
    total_steps += p->w_steps[s2];
  }
  p->nrows = nrows;
  p->c_rounds = 8;
  p->fpb = waves * p->c_rounds * 4;
  p->fpb_unit = waves * 4;
  p->w_waves = waves;
  p->c_rounds_max = 32;
  p->fast_lds_bytes = lds;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64 * waves, lds) == hipSuccess) p->blocks_per_cu = nb;
  char nm[160];
  snprintf(nm, sizeof(nm), "fft1024c_kernel<%d> fbank%s waves=%d lds=%zuB blocks/CU=%d mel4=%dx%d", nrows, fixed ? " fixed-schedule" : "", waves, lds, p->blocks_per_cu, sch.nsets, total_steps);
  p->kernel_name = nm;
  p->variant = 8;
  return HIPFEAT_OK;
}

// --------------------------------------------------------------------------------------
// fft256 fast path (kernel_fft256.hpp): 8 kHz 25/10 ms frames, or <= 16 ms frames at 16 kHz
// --------------------------------------------------------------------------------------
template <int NROWS, int OUT>
static const void* fft256_entry() {
  return reinterpret_cast<const void*>(&fft256_kernel<NROWS, OUT>);
}

// fft256 wave-autonomous log-mel kernel (kernel_fft256c.hpp); shares d_c_shared / c_* with the other wave-autonomous kernels
template <int NROWS, int NFULL>
static const void* fft256c_entry() {
  return reinterpret_cast<const void*>(&fft256c_kernel<NROWS, NFULL>);
}

static hipfeat_status setup_fft256c(hipfeat_plan* p, const float* h_window, const float* h_mel, int nrows) {
  const hipfeat_config& c = p->cfg;
  const int N = c.frame_length, shift = c.frame_shift, M = c.num_filters;
  Mel4Schedule sch;
  if (!build_mel4_schedule(h_mel, M, p->K, kDPRowStride, kDSets, kDSteps, sch)) return HIPFEAT_OK;
  // LDS image: window/2 pairs per (row n1, lane q) | W_128^(q k1) per (row k1, lane q) | split-step twiddles -i W_256^(q + 8 j)
  std::vector<float> img((size_t)(nrows * 8 + 128 + 64) * 2, 0.0f);
  for (int n1 = 0; n1 < nrows; ++n1)
    for (int q = 0; q < 8; ++q)
      for (int e = 0; e < 2; ++e) {
        const int i = 16 * n1 + 2 * q + e;
        img[2 * (n1 * 8 + q) + e] = i < N ? 0.5f * h_window[i] : 0.0f;
      }
  float* twp = img.data() + 2 * nrows * 8;
  float* tws = twp + 256;
  for (int k1 = 0; k1 < 16; ++k1)
    for (int q = 0; q < 8; ++q) {
      const double a = -2.0 * M_PI * (double)(q * k1) / 128.0;
      twp[2 * (k1 * 8 + q)] = (float)std::cos(a);
      twp[2 * (k1 * 8 + q) + 1] = (float)std::sin(a);
    }
  for (int j = 0; j < 8; ++j)
    for (int q = 0; q < 8; ++q) {  // w = -i * W_256^k = (sin(a), -cos(a)) with a = -2 pi k / 256
      const double a = -2.0 * M_PI * (double)(q + 8 * j) / 256.0;
      tws[2 * (j * 8 + q)] = (float)std::sin(a);
      tws[2 * (j * 8 + q) + 1] = (float)(-std::cos(a));
    }
  // the kernel runs kDSets sets of kDSteps steps unconditionally: pad the tables (weights 0, no output column)
  p->c_wtab_off = (int)img.size();
  img.resize(img.size() + (size_t)kDSets * kDSteps * 64, 0.0f);
  for (int s2 = 0; s2 < sch.nsets; ++s2)
    std::memcpy(img.data() + p->c_wtab_off + (size_t)s2 * kDSteps * 64, sch.wtab.data() + (size_t)sch.step0[s2] * 64, (size_t)sch.steps[s2] * 64 * sizeof(float));
  p->c_ltab_off = (int)img.size();
  img.resize(img.size() + (size_t)kDSets * 256, 0.0f);
  {
    const int none = kMel4NoColumn;
    for (int s2 = 0; s2 < kDSets; ++s2)
      for (int lane = 0; lane < 64; ++lane) {
        float* lt = img.data() + p->c_ltab_off + ((size_t)s2 * 64 + lane) * 4;
        if (s2 < sch.nsets) std::memcpy(lt, sch.ltab.data() + ((size_t)s2 * 64 + lane) * 4, 4 * sizeof(float));
        else std::memcpy(lt + 1, &none, 4);
      }
  }
  while (img.size() % 64) img.push_back(0.0f);
  p->c_shared_floats = (int)img.size();
  p->c_xs_floats = (7 * shift + 16 * nrows + 3) & ~3;
  const size_t lds = ((size_t)p->c_shared_floats + (size_t)kDWaves * (p->c_xs_floats + kDRegion)) * sizeof(float);
  if (lds > 80 * 1024 || (p->c_xs_floats >> 8) > 6) return HIPFEAT_OK;  // two workgroups of 8 waves per CU or nothing
  // 25 ms at 8 kHz (N = 200: 12 full rows of 16 samples + a partial one) gets the instance without length masks on the full rows
  const void* fn = nrows == 13 ? (N >= 192 ? fft256c_entry<13, 12>() : fft256c_entry<13, 0>()) : fft256c_entry<16, 0>();
  hipError_t e = ensure_dynamic_lds(fn, lds);
  if (e != hipSuccess) return fail(HIPFEAT_ERR_HIP, "hipFuncSetAttribute(fft256c) failed: %s", hipGetErrorName(e));
  hipfeat_status st;
  if ((st = upload(&p->d_c_shared, img.data(), img.size())) != HIPFEAT_OK) return st;
  p->nrows = nrows;
  p->c_rounds = 4;  // 8 waves x 4 rounds x 8 frames = 256 frames per workgroup
  p->fpb = kDWaves * p->c_rounds * 8;
  p->fpb_unit = kDWaves * 8;
  p->c_rounds_max = 16;
  p->fast_lds_bytes = lds;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64 * kDWaves, lds) == hipSuccess) p->blocks_per_cu = nb;
  int total_steps = 0;
  for (int s2 = 0; s2 < sch.nsets; ++s2) total_steps += sch.steps[s2];
  char nm[128];
  snprintf(nm, sizeof(nm), "fft256c_kernel<%d> fbank lds=%zuB blocks/CU=%d mel4=%dx%d", nrows, lds, p->blocks_per_cu, sch.nsets, total_steps);
  p->kernel_name = nm;
  p->variant = 11;
  return HIPFEAT_OK;
}

static hipfeat_status setup_fft256(hipfeat_plan* p, const float* h_window, const float* h_mel, const float* h_dct, const float* h_lifter) {
  const hipfeat_config& c = p->cfg;
  const int N = c.frame_length, shift = c.frame_shift, M = c.num_filters;
  const bool mfcc = c.kind == HIPFEAT_MFCC;
  const bool spec = c.kind == HIPFEAT_SPECTROGRAM || c.kind == HIPFEAT_LOG_SPECTROGRAM;
  if (c.kind > HIPFEAT_MFCC || c.fft_length != 256 || (shift & 1) || N < 16 || c.use_energy || (!spec && c.use_fft_mag) ||
      route_env("HIPFEAT_FORCE_GENERIC"))
    return HIPFEAT_OK;
  if (mfcc && (M > 8 * kMaxDctGroups || c.num_ceps > 64)) return HIPFEAT_OK;
  const int need = (N + 15) / 16;
  const int nrows = need <= 13 ? 13 : 16;
  const int ntiles = spec ? 0 : (M + 15) / 16;
  if (ntiles > 8) return HIPFEAT_OK;
  if (!mfcc && !spec) {  // log-mel filterbank: the wave-autonomous kernel, unless the schedule or the LDS budget says no
    const char* var = route_env("HIPFEAT_FFT256_VARIANT");
    if (!(var && var[0] == 'b') && !route_env("HIPFEAT_NO_WAVE_AUTONOMOUS")) {  // HIPFEAT_FFT256_VARIANT=b: the 32-frame-tile kernel (tests compare the two)
      hipfeat_status stc = setup_fft256c(p, h_window, h_mel, nrows);
      if (stc != HIPFEAT_OK || p->variant == 11) return stc;
    }
  }
  WaveWork work[4];
  std::vector<float> mel_a((size_t)4 * kMelARegs * 64, 0.0f);
  std::memset(work, 0, sizeof(work));
  if (!spec && !build_mel_schedule(h_mel, M, p->K, k256PRowStride, ntiles, work, mel_a)) return HIPFEAT_OK;
  // LDS constants: window/2 pairs per (row n1, lane q) | W_128^(q k1) per (row k1, lane q) | split-step twiddles
  // w = -i W_256^(q + 8 j) per (row j < 8, lane q) | (-w.y, w.x)
  std::vector<float> wh(256, 0.0f);
  for (int i = 0; i < N; ++i) wh[i] = 0.5f * h_window[i];
  const int const_floats = (nrows * 8 + 128 + 64 + 64) * 2;
  std::vector<float> lc((size_t)const_floats, 0.0f);
  for (int n1 = 0; n1 < nrows; ++n1)
    for (int q = 0; q < 8; ++q) {
      lc[2 * (n1 * 8 + q)] = wh[16 * n1 + 2 * q];
      lc[2 * (n1 * 8 + q) + 1] = wh[16 * n1 + 2 * q + 1];
    }
  float* twp = lc.data() + 2 * nrows * 8;
  float* tws = twp + 256;
  float* twsp = tws + 128;
  for (int k1 = 0; k1 < 16; ++k1)
    for (int q = 0; q < 8; ++q) {
      const double a = -2.0 * M_PI * (double)(q * k1) / 128.0;
      twp[2 * (k1 * 8 + q)] = (float)std::cos(a);
      twp[2 * (k1 * 8 + q) + 1] = (float)std::sin(a);
    }
  for (int j = 0; j < 8; ++j)
    for (int q = 0; q < 8; ++q) {  // w = -i * W_256^k = (sin(a), -cos(a)) with a = -2 pi k / 256
      const double a = -2.0 * M_PI * (double)(q + 8 * j) / 256.0;
      const float wx = (float)std::sin(a), wy = (float)(-std::cos(a));
      tws[2 * (j * 8 + q)] = wx;
      tws[2 * (j * 8 + q) + 1] = wy;
      twsp[2 * (j * 8 + q)] = -wy;
      twsp[2 * (j * 8 + q) + 1] = wx;
    }
  hipfeat_status st;
  if ((st = upload(&p->d_lds_consts, lc.data(), lc.size())) != HIPFEAT_OK) return st;
  if ((st = upload(&p->d_work, work, 4)) != HIPFEAT_OK) return st;
  std::vector<float> mel_a4(mel_a.size());  // weights as 16-byte vectors: [wave][step / 4][lane][step % 4]
  for (int w = 0; w < 4; ++w)
    for (int s4 = 0; s4 < kMelARegs; ++s4)
      for (int lane = 0; lane < 64; ++lane)
        mel_a4[(((size_t)w * kBMelVec + s4 / 4) * 64 + lane) * 4 + (s4 & 3)] = mel_a[((size_t)w * kMelARegs + s4) * 64 + lane];
  if ((st = upload(&p->d_mel_a4, mel_a4.data(), mel_a4.size())) != HIPFEAT_OK) return st;
  p->nrows = nrows;
  p->tiles_per_block = 16;  // 512 frames per workgroup (measured: 8 -> 3.84 M, 16 -> 3.99 M, 32 -> 3.93 M cuts/s at 8 kHz fbank-80)
  p->const_floats = const_floats;
  p->xs_floats = ((k256TileFrames - 1) * shift + 16 * nrows + 255) & ~255;  // whole 1 KiB LDS-DMA chunks
  size_t lds_floats = (size_t)p->xs_floats + const_floats + 4 * k256WaveRegion;
  if (mfcc) {
    p->dct_groups = (M + 7) / 8;
    p->lm_stride = ntiles <= 2 ? 36 : (ntiles <= 4 ? 68 : 132);
    std::vector<float> da = build_dct_operands(c, h_dct, h_lifter, p->dct_groups);
    p->dct_floats = (int)da.size();
    if ((st = upload(&p->d_dct_consts, da.data(), da.size())) != HIPFEAT_OK) return st;
    lds_floats += (size_t)k256TileFrames * p->lm_stride + da.size();
    p->fast_mfcc = true;
  }
  p->fast_lds_bytes = lds_floats * sizeof(float);
  p->fast_out = mfcc ? 1 : (spec ? 2 : 0);
  if (p->fast_lds_bytes > 64 * 1024) return HIPFEAT_OK;  // keep at least two workgroups per CU; otherwise the generic kernel
  const void* fn;
  if (mfcc) fn = nrows == 13 ? fft256_entry<13, 1>() : fft256_entry<16, 1>();
  else if (spec) fn = nrows == 13 ? fft256_entry<13, 2>() : fft256_entry<16, 2>();
  else fn = nrows == 13 ? fft256_entry<13, 0>() : fft256_entry<16, 0>();
  hipError_t e = ensure_dynamic_lds(fn, p->fast_lds_bytes);
  if (e != hipSuccess) return fail(HIPFEAT_ERR_HIP, "hipFuncSetAttribute(fft256) failed: %s", hipGetErrorName(e));
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 256, p->fast_lds_bytes) == hipSuccess) p->blocks_per_cu = nb;
  char nm[112];
  snprintf(nm, sizeof(nm), "fft256_kernel<%d,%d> %s lds=%zuB blocks/CU=%d", nrows, p->fast_out, mfcc ? "mfcc" : (spec ? "spectrogram" : "fbank"),
           p->fast_lds_bytes, p->blocks_per_cu);
  p->kernel_name = nm;
  p->variant = 4;
  p->fpb = k256TileFrames * p->tiles_per_block;
  return HIPFEAT_OK;
}

// --------------------------------------------------------------------------------------
// wave-per-frame kernel (kernel_wave.hpp): power-of-two fft 256 .. 2048 without a specialised kernel
// --------------------------------------------------------------------------------------
template <int N1>
static const void* wave_entry() {
  return reinterpret_cast<const void*>(&wave_kernel<N1>);
}

static hipfeat_status setup_wave(hipfeat_plan* p, const float* h_mel) {
  const hipfeat_config& c = p->cfg;
  const bool librosa = c.kind == HIPFEAT_LIBROSA_FBANK;
  if (p->variant != 0 || !p->pow2 || (c.kind > HIPFEAT_MFCC && !librosa) || route_env("HIPFEAT_FORCE_GENERIC") || route_env("HIPFEAT_NO_WAVE_KERNEL"))
    return HIPFEAT_OK;
  const int H = p->H;
  // H = 128 (fft 256) stays on the radix-2 kernel: measured 0.92 M vs 0.72 M cuts/s there; H = 256: 0.46 vs 0.48 M
  if (!(H == 256 || H == 512 || H == 1024) || c.num_filters > 128) return HIPFEAT_OK;
  const bool need_mel = c.kind == HIPFEAT_FBANK || c.kind == HIPFEAT_MFCC || librosa;
  const int M = need_mel ? c.num_filters : 0;
  hipfeat_status st;
  if (need_mel) {
    int maxband = 1;
    std::vector<int2> rng(M);
    for (int j = 0; j < M; ++j) {
      int lo = p->K, hi = 0;
      for (int k = 0; k < p->K; ++k)
        if (h_mel[(size_t)k * M + j] != 0.0f) {
          lo = std::min(lo, k);
          hi = std::max(hi, k + 1);
        }
      if (hi == 0) lo = 0;
      rng[j] = make_int2(lo, hi);
      maxband = std::max(maxband, hi - lo);
    }
    // blob = [M] int4 {lo rounded down to a multiple of 4, offset of the filter's weights, number of float4 groups, 0} followed
    // by the weights themselves, each filter zero-padded to whole float4 groups
    std::vector<int32_t> desc((size_t)4 * M, 0);
    std::vector<float> wts;
    for (int j = 0; j < M; ++j) {
      const int lo4 = rng[j].x & ~3;
      const int groups = rng[j].y > rng[j].x ? (rng[j].y - lo4 + 3) / 4 : 0;
      desc[(size_t)4 * j] = lo4;
      desc[(size_t)4 * j + 1] = (int32_t)wts.size();
      desc[(size_t)4 * j + 2] = groups;
      for (int t = 0; t < 4 * groups; ++t) {
        const int k = lo4 + t;
        wts.push_back(k < p->K ? h_mel[(size_t)k * M + j] : 0.0f);
      }
    }
    std::vector<float> blob((size_t)4 * M + wts.size());
    std::memcpy(blob.data(), desc.data(), desc.size() * sizeof(int32_t));
    std::memcpy(blob.data() + 4 * M, wts.data(), wts.size() * sizeof(float));
    if ((st = upload(&p->d_mel_t, blob.data(), blob.size())) != HIPFEAT_OK) return st;
    p->mel_maxband = maxband;
    p->wave_blob_floats = (int)blob.size();
  }
  p->wave_dct_in_lds = c.kind == HIPFEAT_MFCC && (size_t)M * c.num_ceps <= 2560;
  auto up4 = [](size_t v) { return (v + 3) & ~(size_t)3; };
  // twiddles W_2H^k + 4 padded wave buffers + window + twiddles W_H^m + filterbank blob (+ DCT matrix)
  p->wave_lds_bytes = ((size_t)2 * H + 4 * ((size_t)144 * (H / 64) + 8) + up4((size_t)c.frame_length) + (size_t)2 * H + up4((size_t)p->wave_blob_floats) +
                       (p->wave_dct_in_lds ? (size_t)M * c.num_ceps : 0)) * sizeof(float);
  const void* fn = H == 256 ? wave_entry<4>() : (H == 512 ? wave_entry<8>() : wave_entry<16>());
  hipError_t e = ensure_dynamic_lds(fn, p->wave_lds_bytes);
  if (e != hipSuccess) return fail(HIPFEAT_ERR_HIP, "hipFuncSetAttribute(wave) failed: %s", hipGetErrorName(e));
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 256, p->wave_lds_bytes) == hipSuccess) p->blocks_per_cu = nb;
  char nm[96];
  snprintf(nm, sizeof(nm), "wave_kernel<%d> fft=%d lds=%zuB blocks/CU=%d", H / 64, c.fft_length, p->wave_lds_bytes, p->blocks_per_cu);
  p->kernel_name = nm;
  p->variant = 5;
  p->fpb = 32;  // 4 waves x 8 frames (the tables copied to LDS per workgroup are ~12 KB)
  return HIPFEAT_OK;
}

// --------------------------------------------------------------------------------------
// whisper FFT fast path: 400 = 16 x 25 mixed-radix FFT on the vector ALUs + banded mel GEMM (kernel_whisper2.hpp)
// --------------------------------------------------------------------------------------
static hipfeat_status setup_whisper2(hipfeat_plan* p, const float* h_mel) {
  const hipfeat_config& c = p->cfg;
  if (c.kind != HIPFEAT_WHISPER || c.frame_length != kW2N || c.frame_shift != kW2Shift || c.num_filters > 16 * kW2MaxMelTiles ||
      route_env("HIPFEAT_FORCE_GENERIC"))
    return HIPFEAT_OK;
  const int M = c.num_filters, nmt = (M + 15) / 16;
  std::vector<float> cs(288);
  for (int j = 1; j <= 12; ++j)
    for (int k = 1; k <= 12; ++k) {
      const double th = 2.0 * M_PI * (double)((j * k) % 25) / 25.0;
      cs[(size_t)(j - 1) * 24 + 2 * (k - 1)] = (float)std::cos(th);
      cs[(size_t)(j - 1) * 24 + 2 * (k - 1) + 1] = (float)-std::sin(th);
    }
  std::vector<float> tw((size_t)13 * 16 * 2);
  for (int k2 = 0; k2 < 13; ++k2)
    for (int l = 0; l < 16; ++l) {
      const double th = 2.0 * M_PI * (double)((l * k2) % 400) / 400.0;
      tw[((size_t)k2 * 16 + l) * 2] = (float)std::cos(th);
      tw[((size_t)k2 * 16 + l) * 2 + 1] = (float)-std::sin(th);
    }
  std::vector<float> mel;
  for (int mt = 0; mt < nmt; ++mt) {
    int lo = 201, hi = 0;
    for (int bin = 0; bin <= 200; ++bin)
      for (int i = 0; i < 16; ++i) {
        const int m = 16 * mt + i;
        if (m < M && h_mel[(size_t)bin * M + m] != 0.0f) {
          lo = std::min(lo, bin);
          hi = std::max(hi, bin + 1);
        }
      }
    if (hi == 0) lo = 0;
    const int k0 = lo & ~3, chunks = hi > lo ? (hi - k0 + 15) / 16 : 0;  // chunks of 4 k-steps (16 bins); zero weights pad the band
    p->wh2_k0[mt] = k0;
    p->wh2_steps[mt] = chunks;
    p->wh2_off[mt] = (int32_t)(mel.size() / 256);
    for (int ch = 0; ch < chunks; ++ch)
      for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
          const int bin = k0 + 16 * ch + 4 * r + (l >> 4), m = 16 * mt + (l & 15);
          mel.push_back((bin <= 200 && m < M) ? h_mel[(size_t)bin * M + m] : 0.0f);
        }
  }
  if (mel.empty()) mel.assign(256, 0.0f);
  // deal the mel tiles to the four waves: longest first, always to the least loaded wave (at most two tiles each)
  int load[4] = {0, 0, 0, 0}, cnt[4] = {0, 0, 0, 0};
  for (int wv = 0; wv < 4; ++wv) p->wh2_wave_tiles[wv][0] = p->wh2_wave_tiles[wv][1] = -1;
  std::vector<int> order(nmt);
  for (int i = 0; i < nmt; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int a, int b) { return p->wh2_steps[a] > p->wh2_steps[b]; });
  for (int mt : order) {
    int best = -1;
    for (int wv = 0; wv < 4; ++wv)
      if (cnt[wv] < 2 && (best < 0 || load[wv] < load[best])) best = wv;
    p->wh2_wave_tiles[best][cnt[best]++] = mt;
    load[best] += p->wh2_steps[mt];
  }
  hipfeat_status st;
  if ((st = upload(&p->d_wh2_cs, cs.data(), cs.size())) != HIPFEAT_OK) return st;
  if ((st = upload(&p->d_wh2_tw, tw.data(), tw.size())) != HIPFEAT_OK) return st;
  if ((st = upload(&p->d_wh2_mel, mel.data(), mel.size())) != HIPFEAT_OK) return st;
  std::vector<int32_t> sched(32, 0);
  for (int wv = 0; wv < 4; ++wv)
    for (int sl = 0; sl < 2; ++sl) {
      const int mt = p->wh2_wave_tiles[wv][sl];
      int32_t* e = &sched[(size_t)(wv * 2 + sl) * 4];
      e[0] = mt;
      if (mt >= 0) e[1] = p->wh2_k0[mt], e[2] = p->wh2_steps[mt], e[3] = p->wh2_off[mt];
    }
  if ((st = upload(&p->d_wh2_sched, sched.data(), sched.size())) != HIPFEAT_OK) return st;
  p->variant = 6;
  p->fpb = 16 * kW2TilesPerBlock;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&whisper2_kernel), 256, 0) == hipSuccess) p->blocks_per_cu = nb;
  char buf[160];
  snprintf(buf, sizeof(buf), "whisper_kernel2 fft400=16x25 mel_chunks=%d+%d+%d+%d blocks/CU=%d", load[0], load[1], load[2], load[3], p->blocks_per_cu);
  p->kernel_name = buf;
  return HIPFEAT_OK;
}

// fft2048 wave-autonomous fbank kernel (kernel_fft2048c.hpp): 44.1 / 48 kHz Kaldi filterbanks, librosa-style log-mel with n_fft 2048
template <int NROWS, bool ODD, int S0 = 0, int S1 = 0, int S2 = 0, bool W12 = false>
static const void* fft2048c_entry() {
  return reinterpret_cast<const void*>(&fft2048c_kernel<NROWS, ODD, S0, S1, S2, W12>);
}

static hipfeat_status setup_fft2048c(hipfeat_plan* p, const float* h_window, const float* h_mel) {
  const hipfeat_config& c = p->cfg;
  const int N = c.frame_length, shift = c.frame_shift, M = c.num_filters;
  const bool librosa = c.kind == HIPFEAT_LIBROSA_FBANK;  // centred frames, |X| or |X|^2, log10 (librosa_fbank.py:66-137)
  if (p->variant != 0 || (c.kind != HIPFEAT_FBANK && !librosa) || c.fft_length != 2048 || N <= 1024 || c.use_energy ||
      (c.use_fft_mag && !librosa) || route_env("HIPFEAT_FORCE_GENERIC") || route_env("HIPFEAT_NO_WAVE_AUTONOMOUS"))
    return HIPFEAT_OK;
  const bool odd = (shift & 1) != 0;
  const int need = (N + 63) / 64;
  const int nrows = odd ? (need <= 18 ? 18 : 32) : (need <= 19 ? 19 : 32);
  Mel4Schedule sch;
  if (!build_mel4_schedule(h_mel, M, p->K, kXPRowStride, kXMaxSets, kXMaxSteps, sch)) return HIPFEAT_OK;
  // LDS image: window/2 as (even, odd) sample pairs per (row n1, lane q)
  std::vector<float> img((size_t)nrows * 32 * 2, 0.0f);
  for (int n1 = 0; n1 < nrows; ++n1)
    for (int q = 0; q < 32; ++q)
      for (int e = 0; e < 2; ++e) {
        const int i = 64 * n1 + 2 * q + e;
        img[2 * (n1 * 32 + q) + e] = i < N ? 0.5f * h_window[i] : 0.0f;
      }
  // split twiddles -i W_2048^k per (step, lane); k = bin of the step's first operand: lanes 1..16: l + 64 s, lanes 17..31:
  // (64 - l) + 64 s; lane 0: 64 s (s <= 8), 32 + 64 (s - 9) (s <= 15), 480 (s = 16)
  p->x_tws_off = (int)img.size();
  img.resize(img.size() + (size_t)kXSplitSteps * 32 * 2, 0.0f);
  for (int st = 0; st < kXSplitSteps; ++st)
    for (int q = 0; q < 32; ++q) {
      int k;
      if (q != 0) k = st < 16 ? (q <= 16 ? q : 64 - q) + 64 * st : 0;
      else k = st <= 8 ? 64 * st : (st <= 15 ? 32 + 64 * (st - 9) : 480);
      const double a = -2.0 * M_PI * (double)k / 2048.0;  // w = -i * W_2048^k = (sin(a), -cos(a))
      img[(size_t)p->x_tws_off + 2 * (st * 32 + q)] = (float)std::sin(a);
      img[(size_t)p->x_tws_off + 2 * (st * 32 + q) + 1] = (float)(-std::cos(a));
    }
  // butterfly twiddles of pass 2: row 0 = ones (even outputs), row 1 = W_32^n (odd outputs)
  p->x_tw32_off = (int)img.size();
  img.resize(img.size() + 2 * 16 * 2, 0.0f);
  for (int n = 0; n < 16; ++n) {
    const double a = -2.0 * M_PI * (double)n / 32.0;
    img[(size_t)p->x_tw32_off + 2 * n] = 1.0f;
    img[(size_t)p->x_tw32_off + 2 * (16 + n)] = (float)std::cos(a);
    img[(size_t)p->x_tw32_off + 2 * (16 + n) + 1] = (float)std::sin(a);
  }
  p->c_wtab_off = (int)img.size();
  img.insert(img.end(), sch.wtab.begin(), sch.wtab.end());
  // a wave carries TWO frames: rows 2 and 3 of every 4 x 4 block read the power rows of frames 0 and 1 again (their results are dropped)
  for (size_t i = 0; i < sch.ltab.size(); i += 4) {
    const int lane = (int)((i / 4) % 64);
    if ((lane & 3) >= 2) {
      int v;
      std::memcpy(&v, &sch.ltab[i], 4);
      v -= 2 * kXPRowStride;
      std::memcpy(&sch.ltab[i], &v, 4);
    }
  }
  p->c_ltab_off = (int)img.size();
  img.insert(img.end(), sch.ltab.begin(), sch.ltab.end());
  while (img.size() % 64) img.push_back(0.0f);
  p->c_shared_floats = (int)img.size();
  p->c_xs_floats = (shift + 64 * nrows + 3) & ~3;
  if ((p->c_xs_floats >> 8) > 10) return HIPFEAT_OK;
  // instance with the mel schedule as compile-time constants (kernel_fft2048c.hpp): the 80-filter Kaldi default at 44.1 / 48 kHz.  The 44.1 kHz one
  // runs 12 waves per workgroup (3 waves/SIMD) without a span prefetch, the span buffer aliasing the exchange / power region (at 48 kHz that
  // layout measured 3 % slower than 8 waves with the prefetch)
  const bool fixed = sch.nsets == 3 && sch.steps[0] == 52 && sch.steps[1] == 28 && sch.steps[2] == 16 && sch.step0[1] == 52 && sch.step0[2] == 80 &&
                     (odd ? nrows == 18 : nrows == 19) && !librosa && !c.use_fft_mag && M == 80 &&
                     (odd ? (N == 1102 && shift == 441) : (N == 1200 && shift == 480)) && !c.snip_edges &&  // the geometry the instances have compiled in (kernel_fft2048c.hpp)
                     !route_env("HIPFEAT_NO_FIXED_SCHEDULE");
  // (HIPFEAT_FFT2048_W12 = 1 / 0: routing switch, forces / forbids the 12-wave layout for either default -- same-call A/Bs)
  const char* w12_env = route_env("HIPFEAT_FFT2048_W12");
  const bool w12_want = w12_env ? w12_env[0] == '1' : odd;
  const bool w12 = fixed && w12_want && p->c_xs_floats <= kXRegion && ((size_t)p->c_shared_floats + (size_t)kXWavesFixed * kXRegion) * sizeof(float) <= 160 * 1024;
  int waves = w12 ? kXWavesFixed : kXMaxWaves;
  auto lds_of = [&](int wv) { return ((size_t)p->c_shared_floats + (size_t)wv * (w12 ? kXRegion : p->c_xs_floats + kXRegion)) * sizeof(float); };
  while (waves > 0 && lds_of(waves) > 160 * 1024) --waves;
  if (waves < 4) return HIPFEAT_OK;
  if (fixed && waves != (w12 ? kXWavesFixed : kXMaxWaves)) return HIPFEAT_OK;  // (cannot happen for the default geometry: its LDS image fits)
  const size_t lds = lds_of(waves);
  const void* fn = fixed ? (odd ? (w12 ? fft2048c_entry<18, true, 52, 28, 16, true>() : fft2048c_entry<18, true, 52, 28, 16>())
                                : (w12 ? fft2048c_entry<19, false, 52, 28, 16, true>() : fft2048c_entry<19, false, 52, 28, 16>()))
                   : odd ? (nrows == 18 ? fft2048c_entry<18, true>() : fft2048c_entry<32, true>())
                         : (nrows == 19 ? fft2048c_entry<19, false>() : fft2048c_entry<32, false>());
  p->w_fixed = fixed ? 1 : 0;
  hipError_t e = ensure_dynamic_lds(fn, lds);
  if (e != hipSuccess) return fail(HIPFEAT_ERR_HIP, "hipFuncSetAttribute(fft2048c) failed: %s", hipGetErrorName(e));
  std::vector<float> twp((size_t)32 * 32 * 2);
  for (int k1 = 0; k1 < 32; ++k1)
    for (int q = 0; q < 32; ++q) {
      const double a = -2.0 * M_PI * (double)(q * k1) / 1024.0;
      twp[2 * ((size_t)k1 * 32 + q)] = (float)std::cos(a);
      twp[2 * ((size_t)k1 * 32 + q) + 1] = (float)std::sin(a);
    }
  hipfeat_status st;
  if ((st = upload(&p->d_c_shared, img.data(), img.size())) != HIPFEAT_OK) return st;
  if ((st = upload(&p->d_x_twp, twp.data(), twp.size())) != HIPFEAT_OK) return st;
  p->w_nsets = sch.nsets;
  int total_steps = 0;
  for (int s2 = 0; s2 < kXMaxSets; ++s2) {
    p->w_steps[s2] = s2 < sch.nsets ? sch.steps[s2] : 0;
    p->w_step0[s2] = s2 < sch.nsets ? sch.step0[s2] : 0;
    total_steps += p->w_steps[s2];
  }
  p->nrows = nrows;
  p->x_odd = odd;
  p->x_waves = waves;
  p->c_rounds = 8;
  p->fpb = waves * p->c_rounds * 2;
  p->fpb_unit = waves * 2;
  p->c_rounds_max = 64;
  p->fast_lds_bytes = lds;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64 * waves, lds) == hipSuccess) p->blocks_per_cu = nb;
  char nm[160];
  snprintf(nm, sizeof(nm), "fft2048c_kernel<%d,%d> fbank%s waves=%d lds=%zuB blocks/CU=%d mel4=%dx%d", nrows, (int)odd, fixed ? " fixed-schedule" : "", waves, lds, p->blocks_per_cu, sch.nsets, total_steps);
  p->kernel_name = nm;
  p->variant = 10;
  return HIPFEAT_OK;
}

// whisper, wave-autonomous with the normalisation fused (kernel_whisper3.hpp): shares the DFT-25 coefficient table of setup_whisper2
template <int NSETS>
static const void* whisper3_entry() {
  return reinterpret_cast<const void*>(&whisper3_kernel<NSETS>);
}

static hipfeat_status setup_whisper3(hipfeat_plan* p, const float* h_window, const float* h_mel) {
  const hipfeat_config& c = p->cfg;
  if (p->variant != 6) return HIPFEAT_OK;  // setup_whisper2 decides whether this is the Whisper fast-path configuration
  const char* v = route_env("HIPFEAT_WHISPER_VARIANT");
  if (v && v[0] == '2') return HIPFEAT_OK;
  const int M = c.num_filters;
  Mel4Schedule sch;
  if (!build_mel4_schedule(h_mel, M, 201, kW3PRowStride, kW3MaxSets, kW3Steps, sch)) return HIPFEAT_OK;
  const int nsets = sch.nsets <= 2 ? 2 : 3;
  std::vector<float> img((size_t)kW3N + 13 * 16 * 2, 0.0f);
  for (int i = 0; i < kW3N; ++i) img[(size_t)i] = h_window[i];
  for (int k2 = 0; k2 < 13; ++k2)
    for (int l = 0; l < 16; ++l) {
      const double th = 2.0 * M_PI * (double)((l * k2) % 400) / 400.0;
      img[(size_t)kW3N + ((size_t)k2 * 16 + l) * 2] = (float)std::cos(th);
      img[(size_t)kW3N + ((size_t)k2 * 16 + l) * 2 + 1] = (float)-std::sin(th);
    }
  // the kernel runs `nsets` sets of kW3Steps steps unconditionally: pad the tables (weights 0, no output column)
  p->c_wtab_off = (int)img.size();
  img.resize(img.size() + (size_t)nsets * kW3Steps * 64, 0.0f);
  for (int s2 = 0; s2 < sch.nsets; ++s2)
    std::memcpy(img.data() + p->c_wtab_off + (size_t)s2 * kW3Steps * 64, sch.wtab.data() + (size_t)sch.step0[s2] * 64, (size_t)sch.steps[s2] * 64 * sizeof(float));
  p->c_ltab_off = (int)img.size();
  img.resize(img.size() + (size_t)nsets * 256, 0.0f);
  {
    const int none = kMel4NoColumn;
    for (int s2 = 0; s2 < nsets; ++s2)
      for (int lane = 0; lane < 64; ++lane) {
        float* lt = img.data() + p->c_ltab_off + ((size_t)s2 * 64 + lane) * 4;
        if (s2 < sch.nsets) std::memcpy(lt, sch.ltab.data() + ((size_t)s2 * 64 + lane) * 4, 4 * sizeof(float));
        else std::memcpy(lt + 1, &none, 4);
      }
  }
  while (img.size() % 64) img.push_back(0.0f);
  const size_t lds = (img.size() + (size_t)kW3Waves * (kW3Span + kW3Region) + kW3Tail) * sizeof(float);
  if (lds > 80 * 1024) return HIPFEAT_OK;  // two workgroups of 8 waves per CU or nothing
  const void* fn = nsets == 2 ? whisper3_entry<2>() : whisper3_entry<3>();
  hipError_t e = ensure_dynamic_lds(fn, lds);
  if (e != hipSuccess) return fail(HIPFEAT_ERR_HIP, "hipFuncSetAttribute(whisper3) failed: %s", hipGetErrorName(e));
  hipfeat_status st;
  if ((st = upload(&p->d_c_shared, img.data(), img.size())) != HIPFEAT_OK) return st;
  p->c_shared_floats = (int)img.size();
  p->w_nsets = nsets;
  p->c_rounds = 8;  // 8 waves x 8 rounds x 4 frames = 256 frames per workgroup
  p->fpb = kW3Waves * p->c_rounds * 4;
  p->fpb_unit = kW3Waves * 4;
  p->c_rounds_max = 16;
  p->fast_lds_bytes = lds;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64 * kW3Waves, lds) == hipSuccess) p->blocks_per_cu = nb;
  int total_steps = 0;
  for (int s2 = 0; s2 < sch.nsets; ++s2) total_steps += sch.steps[s2];
  char nm[160];
  snprintf(nm, sizeof(nm), "whisper3_kernel<%d> fft400=16x25 fused-norm lds=%zuB blocks/CU=%d mel4=%dx%d", nsets, lds, p->blocks_per_cu, sch.nsets, total_steps);
  p->kernel_name = nm;
  p->variant = 9;
  return HIPFEAT_OK;
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_plan_create(const hipfeat_config* cfg, const float* h_window,
                                              const float* h_mel, const float* h_dct, const float* h_lifter,
                                              int32_t device, hipfeat_plan** out) {
  if (!cfg || !out) return fail(HIPFEAT_ERR_INVALID, "cfg/plan pointer is NULL");
  *out = nullptr;
  if (cfg->struct_size != (int32_t)sizeof(hipfeat_config))
    return fail(HIPFEAT_ERR_INVALID, "hipfeat_config.struct_size %d != %zu (ABI mismatch)", cfg->struct_size,
                sizeof(hipfeat_config));
  const int N = cfg->frame_length, shift = cfg->frame_shift, fft = cfg->fft_length;
  if (cfg->kind < 0 || cfg->kind > 5) return fail(HIPFEAT_ERR_INVALID, "unknown kind %d", cfg->kind);
  if (N <= 0 || shift <= 0 || fft < N)
    return fail(HIPFEAT_ERR_INVALID, "need frame_length>0, frame_shift>0, fft_length>=frame_length (got %d, %d, %d)",
                N, shift, fft);
  if (shift > N)
    return fail(HIPFEAT_ERR_UNSUPPORTED, "frame_shift (%d) > frame_length (%d) is not supported", shift, N);
  if (!h_window) return fail(HIPFEAT_ERR_INVALID, "window is NULL");
  if (cfg->dither != 0.0f)
    return fail(HIPFEAT_ERR_UNSUPPORTED, "dither != 0 is not supported by the library (ABI v%d): the host adds it to the samples", HIPFEAT_ABI_VERSION);
  const bool whisper = cfg->kind == HIPFEAT_WHISPER;
  if (whisper && (fft != N || cfg->snip_edges || cfg->use_energy || cfg->use_fft_mag || cfg->remove_dc_offset || cfg->preemph_coeff != 0.0f))
    return fail(HIPFEAT_ERR_INVALID, "whisper: needs fft_length == frame_length and no snip_edges / energy / magnitude / DC removal / pre-emphasis");
  const bool librosa = cfg->kind == HIPFEAT_LIBROSA_FBANK;
  if (librosa && (cfg->snip_edges || cfg->use_energy))
    return fail(HIPFEAT_ERR_INVALID, "librosa fbank: snip_edges / use_energy are not defined");
  const bool need_mel = cfg->kind == HIPFEAT_FBANK || cfg->kind == HIPFEAT_MFCC || whisper || librosa;
  if (need_mel && (cfg->num_filters <= 0 || !h_mel))
    return fail(HIPFEAT_ERR_INVALID, "fbank/mfcc need num_filters>0 and a mel matrix");
  if (cfg->kind == HIPFEAT_MFCC) {
    if (cfg->num_ceps <= 0 || !h_dct) return fail(HIPFEAT_ERR_INVALID, "mfcc needs num_ceps>0 and a dct matrix");
    if (cfg->apply_lifter && !h_lifter) return fail(HIPFEAT_ERR_INVALID, "apply_lifter set but lifter is NULL");
  }
  if (fft > 8192) return fail(HIPFEAT_ERR_UNSUPPORTED, "fft_length %d > 8192 is not supported", fft);

  hipfeat_plan* p = new (std::nothrow) hipfeat_plan();
  if (!p) return fail(HIPFEAT_ERR_INVALID, "out of host memory");
  p->cfg = *cfg;
  p->device = device;
  p->K = fft / 2 + 1;
  p->pow2 = (fft & (fft - 1)) == 0 && fft >= 2;
  p->H = p->pow2 ? fft / 2 : 0;
  p->log2H = 0;
  while (p->pow2 && (1 << p->log2H) < p->H) ++p->log2H;
  p->npad_left = (whisper || librosa) ? N / 2 : (cfg->snip_edges ? 0 : (N - shift) / 2);
  const int M = need_mel ? cfg->num_filters : 0;
  const int C = cfg->kind == HIPFEAT_MFCC ? cfg->num_ceps : 0;
  p->feature_dim = cfg->kind == HIPFEAT_FBANK ? M + (cfg->use_energy ? 1 : 0) : (cfg->kind == HIPFEAT_MFCC ? C : ((whisper || librosa) ? M : p->K));

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    delete p;
    return fail(HIPFEAT_ERR_HIP, "device %d not available (%d HIP devices visible)", device, ndev);
  }
  {  // the kernels are gfx950 code objects and lean on gfx950's memory pipeline (hipfeat.h, HIPFEAT_WHISPER): any other device is refused here
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
      delete p;
      return fail(HIPFEAT_ERR_UNSUPPORTED, "device %d is not a gfx950 (MI355X-class) GPU: libhipfeat is built for that architecture alone", device);
    }
  }
  DeviceGuard g(device);
  hipfeat_status st = HIPFEAT_OK;
  auto bail = [&](hipfeat_status s) {
    plan_free(p);
    return s;
  };

  // twiddles, computed in double: W_fft^k = exp(-2 pi i k / fft)
  {
    const int nt = p->pow2 ? std::max(p->H, 1) : fft;
    std::vector<float2> tw(nt);
    for (int k = 0; k < nt; ++k) {
      const double a = -2.0 * M_PI * (double)k / (double)fft;
      tw[k] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    if ((st = upload(&p->d_tw, tw.data(), tw.size())) != HIPFEAT_OK) return bail(st);
  }
  if ((st = upload(&p->d_window, h_window, (size_t)N)) != HIPFEAT_OK) return bail(st);
  if (need_mel) {
    if ((st = upload(&p->d_mel, h_mel, (size_t)p->K * M)) != HIPFEAT_OK) return bail(st);
    // band of each filter: [first non-zero bin, last non-zero bin + 1)
    std::vector<int2> rng(M);
    for (int j = 0; j < M; ++j) {
      int lo = p->K, hi = 0;
      for (int k = 0; k < p->K; ++k)
        if (h_mel[(size_t)k * M + j] != 0.0f) {
          lo = std::min(lo, k);
          hi = std::max(hi, k + 1);
        }
      if (hi == 0) lo = 0;
      rng[j] = make_int2(lo, hi);
    }
    if ((st = upload(&p->d_mel_range, rng.data(), rng.size())) != HIPFEAT_OK) return bail(st);
  }
  if (C) {
    if ((st = upload(&p->d_dct, h_dct, (size_t)M * C)) != HIPFEAT_OK) return bail(st);
    if (cfg->apply_lifter && (st = upload(&p->d_lifter, h_lifter, (size_t)C)) != HIPFEAT_OK) return bail(st);
  }

  // LDS carve-up of the generic kernel.  Frames per workgroup: the largest of 8/4/2 whose footprint stays <= 21 KB
  // (7-8 workgroups = full wave occupancy per CU); large FFTs that cannot get there take 2 frames if that fits 64 KB,
  // else 1.  Measured on MI355X (cuts/s of 10 s cuts): fft 512: 8 -> 4 frames 373 k -> 458 k; fft 1024: 8 -> 2 frames
  // 117 k -> 209 k; fft 2048: 8 -> 2 frames 30 k -> 67 k (1 frame: 59 k).
  auto carve = [&](int fpb) {
    auto al = [](int v) { return (v + 3) & ~3; };
    p->fpb = fpb;
    p->span = (fpb - 1) * shift + N;
    p->off_z = al(p->span);
    p->off_p = p->off_z + al(fpb * fft);
    p->off_tw = p->off_p + al(fpb * p->K);
    p->off_stat = p->off_tw + (p->pow2 ? al(2 * std::max(p->H, 1)) : 0);
    p->off_mel = p->off_stat + al(2 * fpb);
    const int end = p->off_mel + al(fpb * std::max(M, 1));
    p->lds_bytes = (size_t)end * sizeof(float);
    return p->lds_bytes;
  };
  {
    size_t cap = 21 * 1024;
    int pick = 0;
    for (int fpb = 8; fpb >= 2 && !pick; fpb >>= 1)
      if (carve(fpb) <= cap) pick = fpb;
    if (!pick) pick = carve(2) <= 64 * 1024 ? 2 : 1;
    if (carve(pick) > 160 * 1024) p->fpb = 0;
  }
  if (p->fpb < 1) return bail(fail(HIPFEAT_ERR_UNSUPPORTED, "configuration does not fit in LDS"));
  hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&generic_kernel), p->lds_bytes);
  if (e != hipSuccess) return bail(fail(HIPFEAT_ERR_HIP, "hipFuncSetAttribute(LDS=%zu) failed: %s", p->lds_bytes, hipGetErrorName(e)));

  st = setup_fft512(p, h_window, h_mel, h_dct, h_lifter);
  if (st != HIPFEAT_OK) return bail(st);
  st = setup_whisper2(p, h_mel);
  if (st != HIPFEAT_OK) return bail(st);
  st = setup_whisper3(p, h_window, h_mel);
  if (st != HIPFEAT_OK) return bail(st);
  st = setup_fft256(p, h_window, h_mel, h_dct, h_lifter);
  if (st != HIPFEAT_OK) return bail(st);
  st = setup_fft1024c(p, h_window, h_mel);
  if (st != HIPFEAT_OK) return bail(st);
  st = setup_fft2048c(p, h_window, h_mel);
  if (st != HIPFEAT_OK) return bail(st);
  st = setup_wave(p, h_mel);
  if (st != HIPFEAT_OK) return bail(st);

  *out = p;
  return HIPFEAT_OK;
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_plan_destroy(hipfeat_plan* plan) {
  plan_free(plan);
  return HIPFEAT_OK;
}

extern "C" HIPFEAT_API int32_t hipfeat_plan_feature_dim(const hipfeat_plan* plan) { return plan ? plan->feature_dim : 0; }
extern "C" HIPFEAT_API const char* hipfeat_plan_kernel_name(const hipfeat_plan* plan) { return plan ? plan->kernel_name.c_str() : ""; }

// --------------------------------------------------------------------------------------
// layout
// --------------------------------------------------------------------------------------
// Validates the batch and fills the host-side descriptor table.
static hipfeat_status build_descs(const hipfeat_plan* plan, int64_t batch, const int64_t* offs, const int64_t* ns,
                                  const int64_t* padded, const int64_t* out_rows, int64_t out_row_stride,
                                  std::vector<CutDesc>& descs, hipfeat_layout* lay) {
  if (!plan) return fail(HIPFEAT_ERR_INVALID, "plan is NULL");
  if (batch < 0 || (batch > 0 && (!offs || !ns))) return fail(HIPFEAT_ERR_INVALID, "bad batch arguments");
  if (out_row_stride < plan->feature_dim)
    return fail(HIPFEAT_ERR_INVALID, "out_row_stride %lld < feature_dim %d", (long long)out_row_stride, plan->feature_dim);
  const hipfeat_config& c = plan->cfg;
  descs.resize((size_t)batch);
  lay->num_frames.resize((size_t)batch);
  int64_t row = 0, blocks = 0;
  int uniform = -1;
  for (int64_t b = 0; b < batch; ++b) {
    const int64_t S = ns[b], P = padded ? padded[b] : ns[b];
    if (S < 0 || P < S || P > INT32_MAX)
      return fail(HIPFEAT_ERR_INVALID, "cut %lld: num_samples=%lld padded_len=%lld out of range", (long long)b, (long long)S, (long long)P);
    int64_t T = hipfeat_num_frames(S, c.frame_length, c.frame_shift, c.snip_edges);
    if (c.kind == HIPFEAT_WHISPER || c.kind == HIPFEAT_LIBROSA_FBANK) {
      if (P != S) return fail(HIPFEAT_ERR_INVALID, "centred framing: zero-padded batch rows (padded_len != num_samples) are not defined");
      if (S <= c.frame_length / 2)  // torch.stft: reflect padding must be shorter than the signal
        return fail(HIPFEAT_ERR_TOO_SHORT, "cut %lld: %lld samples are not longer than the reflect padding (%d)", (long long)b, (long long)S,
                    c.frame_length / 2);
    } else if (padded) {
      // _extract_batch (extractors.py:499-537): the padded row of P samples is framed as a whole and
      // item b keeps the first compute_num_frames_from_samples(S) rows (lhotse/utils.py:424-434) --
      // with snip_edges that is NOT the snip_edges count, and it is capped by what the row yields.
      const int64_t hop = c.batch_hop > 0 ? c.batch_hop : c.frame_shift;
      T = std::min<int64_t>((S + hop / 2) / hop, hipfeat_num_frames(P, c.frame_length, c.frame_shift, c.snip_edges));
    }
    if (!c.snip_edges && T > 0 && c.kind != HIPFEAT_WHISPER && c.kind != HIPFEAT_LIBROSA_FBANK) {
      hipfeat_status st = hipfeat_check_length(P, c.frame_length, c.frame_shift, 0);
      if (st != HIPFEAT_OK) return st;
    }
    if (!c.snip_edges && T == 0)
      return fail(HIPFEAT_ERR_TOO_SHORT, "cut %lld: %lld samples yield no frames", (long long)b, (long long)S);
    CutDesc& d = descs[(size_t)b];
    d.wave_off = offs[b];
    d.out_row = out_rows ? out_rows[b] : row;
    d.num_samples = (int32_t)S;
    d.padded_len = (int32_t)P;
    d.num_frames = (int32_t)T;
    row += T;
    lay->num_frames[(size_t)b] = T;
  }
  // frames per workgroup: fixed by the plan, or (wave-autonomous kernels) rounds x fpb_unit.  A workgroup pays a fixed start-up (the
  // constant image, its first, un-overlapped span: ~0.64 of a round, from the 8-vs-16-rounds A/B of round 3), its last round-set is
  // only partly filled (a cut's frames are not shared between workgroups), and the launch runs in ceil(workgroups / resident slots)
  // waves of workgroups: the rounds that minimise  waves x (start-up + rounds).  10 000 x 1000 frames end up at the maximum (16: two
  // workgroups per cut, start-up amortised); LibriSpeech-like lengths (mean 1230 frames) at 8 (16 would leave the third workgroup of a
  // cut 60 % empty); a 600 s mini-batch (60 000 frames) at 4 rounds in ONE wave of ~470 workgroups instead of two waves of 2-round
  // workgroups (round 3's rule: the largest power of two that still gave four waves, else 2).  Evaluated on at most 512 evenly spaced
  // cuts of the batch (a transient layout is built per call).
  int fpb = plan->fpb;
  if (plan->fpb_unit > 0) {
    static const int forced = exp_env("HIPFEAT_ROUNDS") ? atoi(exp_env("HIPFEAT_ROUNDS")) : 0;
    static const bool old_rule = exp_env("HIPFEAT_ROUNDS_R3") != nullptr;
    const int64_t slots = 256LL * std::max(plan->blocks_per_cu, 1);
    const int64_t stride = std::max<int64_t>(1, batch / 512);
    auto workgroups = [&](int rounds, int64_t step) {  // (estimate for step > 1)
      const int64_t per = (int64_t)plan->fpb_unit * rounds;
      int64_t nb = 0, n = 0;
      for (int64_t b = 0; b < batch; b += step, ++n) nb += (lay->num_frames[(size_t)b] + per - 1) / per;
      return step == 1 ? nb : (nb * batch + n / 2) / std::max<int64_t>(n, 1);
    };
    int rounds = plan->c_rounds_max;
    if (forced >= 1 && forced <= plan->c_rounds_max) {
      rounds = forced;
    } else if (old_rule) {
      for (; rounds > 2; rounds >>= 1)
        if (workgroups(rounds, 1) >= 4 * slots) break;
    } else {
      double best = -1.0;
      for (int r = std::min(2, plan->c_rounds_max); r <= plan->c_rounds_max; ++r) {
        const int64_t nb = workgroups(r, stride);
        const double waves = nb >= 8 * slots ? (double)nb / (double)slots : (double)((nb + slots - 1) / slots);  // (many waves: the last one hardly matters)
        const double cost = waves * (0.64 + r);
        if (best < 0.0 || cost <= best * (1.0 + 1e-9)) {  // ties go to the larger workgroup
          best = cost;
          rounds = r;
        }
      }
    }
    fpb = plan->fpb_unit * rounds;
  }
  // Ragged batch on the 16 kHz default kernel: laid out by frame quads instead of by cuts (kernel_fft512c.hpp, FLAT) -- no half-empty last
  // workgroup per cut, long workgroups whatever the cuts' lengths
  bool ragged = false;
  for (int64_t b = 1; b < batch && !ragged; ++b) ragged = lay->num_frames[(size_t)b] != lay->num_frames[0];
  static const bool no_flat = route_env("HIPFEAT_NO_FLAT") != nullptr;
  lay->flat = ragged && !no_flat && plan->variant == 7 && fft512c_has_flat(plan->nrows, c.frame_length);
  lay->total_quads = 0;
  lay->block_cut.clear();
  if (lay->flat) {
    int64_t quads = 0;
    for (int64_t b = 0; b < batch; ++b) {
      if (quads > INT32_MAX - (1 << 24)) return fail(HIPFEAT_ERR_INVALID, "batch too large for one launch");
      descs[(size_t)b].first_block = (int32_t)quads;
      quads += (lay->num_frames[(size_t)b] + 3) / 4;
    }
    static const int forced = exp_env("HIPFEAT_ROUNDS") ? atoi(exp_env("HIPFEAT_ROUNDS")) : 0;
    const int64_t slots = 256LL * std::max(plan->blocks_per_cu, 1);
    const int waves_per_wg = plan->fpb_unit / 4;  // a wave takes one quad per round
    int rounds = plan->c_rounds_max;
    if (forced >= 1 && forced <= plan->c_rounds_max) {
      rounds = forced;
    } else {
      double best = -1.0;
      for (int r = std::min(2, plan->c_rounds_max); r <= plan->c_rounds_max; ++r) {
        const int64_t nb = (quads + (int64_t)waves_per_wg * r - 1) / ((int64_t)waves_per_wg * r);
        const double waves = nb >= 8 * slots ? (double)nb / (double)slots : (double)((nb + slots - 1) / slots);
        const double cost = waves * (0.64 + r);
        if (best < 0.0 || cost <= best * (1.0 + 1e-9)) {
          best = cost;
          rounds = r;
        }
      }
    }
    fpb = plan->fpb_unit * rounds;
    const int64_t qpw = (int64_t)waves_per_wg * rounds;
    blocks = (quads + qpw - 1) / qpw;
    lay->total_quads = quads;
    lay->block_cut.resize((size_t)blocks);
    int64_t cutp = 0;
    for (int64_t g = 0; g < blocks; ++g) {  // the cut that holds a workgroup's first quad
      const int64_t q = g * qpw;
      while (cutp + 1 < batch && descs[(size_t)cutp + 1].first_block <= q) ++cutp;
      lay->block_cut[(size_t)g] = (int32_t)cutp;
    }
    uniform = 0;
  } else {
    for (int64_t b = 0; b < batch; ++b) {
      if (blocks > INT32_MAX - (1 << 24)) return fail(HIPFEAT_ERR_INVALID, "batch too large for one launch");
      descs[(size_t)b].first_block = (int32_t)blocks;
      const int64_t nb = (lay->num_frames[(size_t)b] + fpb - 1) / fpb;
      blocks += nb;
      if (uniform == -1) uniform = (int)nb;
      else if (uniform != (int)nb) uniform = 0;
    }
  }
  lay->batch = batch;
  lay->total_frames = row;
  lay->total_blocks = blocks;
  lay->out_row_stride = out_row_stride;
  lay->uniform_bpc = uniform > 0 ? uniform : -1;  // -1: ragged, the workgroup -> cut map sits behind the descriptor table (common.hpp)
  if (uniform <= 0 && !lay->flat) {
    lay->block_cut.resize((size_t)blocks);
    for (int64_t b = 0; b < batch; ++b) {
      const int64_t b0 = descs[(size_t)b].first_block, b1 = b + 1 < batch ? descs[(size_t)b + 1].first_block : blocks;
      std::fill(lay->block_cut.begin() + b0, lay->block_cut.begin() + b1, (int32_t)b);
    }
  }
  lay->fpb = fpb;
  lay->fpb_unit = plan->fpb_unit;
  lay->device = plan->device;
  return HIPFEAT_OK;
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_layout_create(const hipfeat_plan* plan, int64_t batch, const int64_t* h_wave_offsets,
                                                const int64_t* h_num_samples, const int64_t* h_padded_len,
                                                const int64_t* h_out_rows, int64_t out_row_stride, void* stream,
                                                hipfeat_layout** layout) {
  if (!layout) return fail(HIPFEAT_ERR_INVALID, "layout pointer is NULL");
  *layout = nullptr;
  hipfeat_layout* lay = new (std::nothrow) hipfeat_layout();
  if (!lay) return fail(HIPFEAT_ERR_INVALID, "out of host memory");
  std::vector<CutDesc> descs;
  hipfeat_status st = build_descs(plan, batch, h_wave_offsets, h_num_samples, h_padded_len, h_out_rows, out_row_stride, descs, lay);
  if (st != HIPFEAT_OK) {
    delete lay;
    return st;
  }
  DeviceGuard g(plan->device);
  // one allocation: descriptors, then (Whisper with the fused normalisation) kNormSlots armed copies of the normalisation scratch
  // (ragged batches: the workgroup -> cut map directly behind the descriptors, common.hpp::block_cut_map)
  const size_t map_bytes = (lay->block_cut.size() * sizeof(int32_t) + 15) & ~(size_t)15;
  const size_t desc_bytes = std::max<size_t>(descs.size(), 1) * sizeof(CutDesc) + map_bytes;
  const size_t norm_bytes = plan->variant == 9 ? (size_t)kNormSlots * norm_slot_bytes(lay) : 0;
  std::vector<unsigned char> blob(desc_bytes + norm_bytes, 0);  // zeros = armed counters
  if (!descs.empty()) std::memcpy(blob.data(), descs.data(), descs.size() * sizeof(CutDesc));
  if (map_bytes) std::memcpy(blob.data() + descs.size() * sizeof(CutDesc), lay->block_cut.data(), lay->block_cut.size() * sizeof(int32_t));
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&lay->d_cuts), blob.size());
  if (e == hipSuccess && !descs.empty()) {
    // synchronous w.r.t. the host (pageable source), ordered on `stream`
    e = hipMemcpyAsync(lay->d_cuts, blob.data(), blob.size(), hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  }
  if (norm_bytes) {
    lay->d_norm = reinterpret_cast<unsigned char*>(lay->d_cuts) + desc_bytes;
    lay->norm_slots = kNormSlots;
  }
  if (e != hipSuccess) {
    (void)hipFree(lay->d_cuts);
    delete lay;
    return fail(HIPFEAT_ERR_HIP, "layout upload failed: %s", hipGetErrorName(e));
  }
  *layout = lay;
  return HIPFEAT_OK;
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_layout_destroy(hipfeat_layout* layout) {
  if (!layout) return HIPFEAT_OK;
  if (layout->owns && layout->d_cuts) {
    DeviceGuard g(layout->device);
    (void)hipFree(layout->d_cuts);
  }
  delete layout;
  return HIPFEAT_OK;
}

extern "C" HIPFEAT_API int64_t hipfeat_layout_total_frames(const hipfeat_layout* layout) { return layout ? layout->total_frames : 0; }

extern "C" HIPFEAT_API hipfeat_status hipfeat_layout_num_frames(const hipfeat_layout* layout, int64_t* h_num_frames) {
  if (!layout || !h_num_frames) return fail(HIPFEAT_ERR_INVALID, "NULL argument");
  std::copy(layout->num_frames.begin(), layout->num_frames.end(), h_num_frames);
  return HIPFEAT_OK;
}

// --------------------------------------------------------------------------------------
// launch
// --------------------------------------------------------------------------------------
static hipfeat_status launch(const hipfeat_plan* plan, const hipfeat_layout* lay, const float* d_wave, float* d_out,
                             hipStream_t stream) {
  if (lay->total_blocks == 0) return HIPFEAT_OK;
  if (!d_wave || !d_out) return fail(HIPFEAT_ERR_INVALID, "wave/out pointer is NULL");
  if ((plan->fpb_unit > 0 ? (lay->fpb_unit != plan->fpb_unit || lay->fpb % plan->fpb_unit != 0) : lay->fpb != plan->fpb) || lay->device != plan->device)
    return fail(HIPFEAT_ERR_INVALID, "layout was created for a different plan");
  const hipfeat_config& c = plan->cfg;
  if (plan->variant == 5) {
    WaveParams wp{};
    wp.wave = d_wave;
    wp.out = d_out;
    wp.cuts = lay->d_cuts;
    wp.window = plan->d_window;
    wp.tw = plan->d_tw;
    wp.mel_blob = plan->d_mel_t;
    wp.mel_blob_floats = plan->wave_blob_floats;
    wp.dct_in_lds = plan->wave_dct_in_lds ? 1 : 0;
    wp.dct = plan->d_dct;
    wp.lifter = plan->d_lifter;
    wp.out_stride = lay->out_row_stride;
    wp.num_cuts = (int32_t)lay->batch;
    wp.uniform_bpc = lay->uniform_bpc;
    wp.frames_per_wave = plan->fpb / 4;
    wp.N = c.frame_length;
    wp.shift = c.frame_shift;
    wp.H = plan->H;
    wp.K = plan->K;
    wp.M = c.num_filters;
    wp.C = c.num_ceps;
    const bool librosa = c.kind == HIPFEAT_LIBROSA_FBANK;
    wp.kind = librosa ? (int)HIPFEAT_FBANK : c.kind;
    wp.flags = (c.remove_dc_offset ? F_REMOVE_DC : 0) | (c.use_energy ? F_USE_ENERGY : 0) | (c.raw_energy ? F_RAW_ENERGY : 0) |
               (c.use_fft_mag ? F_FFT_MAG : 0) | (c.apply_lifter ? F_LIFTER : 0) | (librosa ? (F_CENTER | F_LOG10) : 0);
    wp.npad_left = plan->npad_left;
    wp.preemph = c.preemph_coeff;
    wp.log_energy_floor = c.energy_floor > 0.0f ? logf(c.energy_floor) : -INFINITY;
    wp.mel_floor = c.mel_floor;
    wp.log_offset = c.log_offset;
    DeviceGuard g(plan->device);
    const dim3 grid((unsigned)lay->total_blocks), block(256);
    set_lds_poison(plan->wave_lds_bytes);
    switch (plan->H >> 6) {
      case 4: hipLaunchKernelGGL(wave_kernel<4>, grid, block, plan->wave_lds_bytes, stream, wp); break;
      case 8: hipLaunchKernelGGL(wave_kernel<8>, grid, block, plan->wave_lds_bytes, stream, wp); break;
      default: hipLaunchKernelGGL(wave_kernel<16>, grid, block, plan->wave_lds_bytes, stream, wp); break;
    }
    HIP_TRY(hipGetLastError());
    return HIPFEAT_OK;
  }
  if (plan->variant == 9) {
    Whisper3Params wp{};
    wp.wave = d_wave;
    wp.out = d_out;
    wp.cuts = lay->d_cuts;
    wp.shared_consts = plan->d_c_shared;
    wp.cs = plan->d_wh2_cs;
    wp.out_stride = lay->out_row_stride;
    wp.num_cuts = (int32_t)lay->batch;
    wp.uniform_bpc = lay->uniform_bpc;
    wp.total_blocks = (int32_t)lay->total_blocks;
    wp.frames_per_block = lay->fpb;
    wp.rounds = lay->fpb / plan->fpb_unit;
    wp.M = c.num_filters;
    wp.mel_floor = c.mel_floor;
    wp.shared_floats = plan->c_shared_floats;
    wp.wtab_off = plan->c_wtab_off;
    wp.ltab_off = plan->c_ltab_off;
    if (lay->d_norm) {
      unsigned char* slot = lay->d_norm + (size_t)(lay->norm_next.fetch_add(1u) % (unsigned)lay->norm_slots) * norm_slot_bytes(lay);
      wp.wg_stat = reinterpret_cast<float*>(slot);
      wp.cut_done = reinterpret_cast<uint32_t*>(slot + 2 * (size_t)lay->total_blocks * sizeof(float));
    }
    DeviceGuard g(plan->device);
    const dim3 grid((unsigned)lay->total_blocks), block(64 * kW3Waves);
    set_lds_poison(plan->fast_lds_bytes);
    if (plan->w_nsets == 2) hipLaunchKernelGGL(whisper3_kernel<2>, grid, block, plan->fast_lds_bytes, stream, wp);
    else hipLaunchKernelGGL(whisper3_kernel<3>, grid, block, plan->fast_lds_bytes, stream, wp);
    HIP_TRY(hipGetLastError());
    if (!wp.wg_stat) {  // no scratch behind this layout: finish with the separate pass
      hipLaunchKernelGGL(whisper_norm_kernel, dim3((unsigned)lay->batch), dim3(1024), 0, stream, lay->d_cuts, d_out, lay->out_row_stride,
                         (int32_t)c.num_filters, (int32_t)c.frame_shift);
      HIP_TRY(hipGetLastError());
    }
    return HIPFEAT_OK;
  }
  if (plan->variant == 6) {
    Whisper2Params wp{};
    wp.wave = d_wave;
    wp.out = d_out;
    wp.cuts = lay->d_cuts;
    wp.window = plan->d_window;
    wp.cs = plan->d_wh2_cs;
    wp.tw = plan->d_wh2_tw;
    wp.mel_a = plan->d_wh2_mel;
    wp.out_stride = lay->out_row_stride;
    wp.num_cuts = (int32_t)lay->batch;
    wp.uniform_bpc = lay->uniform_bpc;
    wp.M = c.num_filters;
    wp.mel_floor = c.mel_floor;
    wp.sched = plan->d_wh2_sched;
    DeviceGuard g(plan->device);
    hipLaunchKernelGGL(whisper2_kernel, dim3((unsigned)lay->total_blocks), dim3(256), 0, stream, wp);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(whisper_norm_kernel, dim3((unsigned)lay->batch), dim3(1024), 0, stream, lay->d_cuts, d_out, lay->out_row_stride,
                       (int32_t)c.num_filters, (int32_t)c.frame_shift);
    HIP_TRY(hipGetLastError());
    return HIPFEAT_OK;
  }
  if (plan->variant == 11) {
    Fft512cParams fp{};
    fp.wave = d_wave;
    fp.out = d_out;
    fp.cuts = lay->d_cuts;
    fp.shared_consts = plan->d_c_shared;
    fp.out_stride = lay->out_row_stride;
    fp.num_cuts = (int32_t)lay->batch;
    fp.uniform_bpc = lay->uniform_bpc;
    fp.frames_per_block = lay->fpb;
    fp.rounds = lay->fpb / plan->fpb_unit;
    fp.N = c.frame_length;
    fp.shift = c.frame_shift;
    fp.npad_left = plan->npad_left;
    fp.M = c.num_filters;
    fp.flags = c.remove_dc_offset ? F_REMOVE_DC : 0;
    fp.preemph = c.preemph_coeff;
    fp.mel_floor = c.mel_floor;
    fp.shared_floats = plan->c_shared_floats;
    fp.wtab_off = plan->c_wtab_off;
    fp.ltab_off = plan->c_ltab_off;
    fp.xs_floats = plan->c_xs_floats;
    DeviceGuard g(plan->device);
    const dim3 grid((unsigned)lay->total_blocks), block(64 * kDWaves);
    set_lds_poison(plan->fast_lds_bytes);
    if (plan->nrows == 13 && c.frame_length >= 192) hipLaunchKernelGGL((fft256c_kernel<13, 12>), grid, block, plan->fast_lds_bytes, stream, fp);
    else if (plan->nrows == 13) hipLaunchKernelGGL((fft256c_kernel<13, 0>), grid, block, plan->fast_lds_bytes, stream, fp);
    else hipLaunchKernelGGL((fft256c_kernel<16, 0>), grid, block, plan->fast_lds_bytes, stream, fp);
    HIP_TRY(hipGetLastError());
    return HIPFEAT_OK;
  }
  if (plan->variant == 10) {
    Fft2048cParams fp{};
    fp.wave = d_wave;
    fp.out = d_out;
    fp.cuts = lay->d_cuts;
    fp.shared_consts = plan->d_c_shared;
    fp.twp = plan->d_x_twp;
    fp.out_stride = lay->out_row_stride;
    fp.num_cuts = (int32_t)lay->batch;
    fp.uniform_bpc = lay->uniform_bpc;
    fp.frames_per_block = lay->fpb;
    fp.rounds = lay->fpb / plan->fpb_unit;
    fp.waves = plan->x_waves;
    fp.N = c.frame_length;
    fp.shift = c.frame_shift;
    fp.npad_left = plan->npad_left;
    fp.M = c.num_filters;
    fp.flags = (c.remove_dc_offset ? F_REMOVE_DC : 0) | (c.use_fft_mag ? F_FFT_MAG : 0) | (c.kind == HIPFEAT_LIBROSA_FBANK ? (F_CENTER | F_LOG10) : 0);
    fp.preemph = c.preemph_coeff;
    fp.mel_floor = c.mel_floor;
    fp.shared_floats = plan->c_shared_floats;
    fp.tws_off = plan->x_tws_off;
    fp.tw32_off = plan->x_tw32_off;
    fp.wtab_off = plan->c_wtab_off;
    fp.ltab_off = plan->c_ltab_off;
    fp.xs_floats = plan->c_xs_floats;
    fp.nsets = plan->w_nsets;
    for (int s2 = 0; s2 < kXMaxSets; ++s2) fp.steps[s2] = plan->w_steps[s2], fp.step0[s2] = plan->w_step0[s2];
    DeviceGuard g(plan->device);
    const dim3 grid((unsigned)lay->total_blocks), block(64 * plan->x_waves);
    set_lds_poison(plan->fast_lds_bytes);
    if (plan->x_odd) {
      if (plan->w_fixed && plan->x_waves == kXWavesFixed) hipLaunchKernelGGL((fft2048c_kernel<18, true, 52, 28, 16, true>), grid, block, plan->fast_lds_bytes, stream, fp);
      else if (plan->w_fixed) hipLaunchKernelGGL((fft2048c_kernel<18, true, 52, 28, 16>), grid, block, plan->fast_lds_bytes, stream, fp);
      else if (plan->nrows == 18) hipLaunchKernelGGL((fft2048c_kernel<18, true>), grid, block, plan->fast_lds_bytes, stream, fp);
      else hipLaunchKernelGGL((fft2048c_kernel<32, true>), grid, block, plan->fast_lds_bytes, stream, fp);
    } else {
      if (plan->w_fixed && plan->x_waves == kXWavesFixed) hipLaunchKernelGGL((fft2048c_kernel<19, false, 52, 28, 16, true>), grid, block, plan->fast_lds_bytes, stream, fp);
      else if (plan->w_fixed) hipLaunchKernelGGL((fft2048c_kernel<19, false, 52, 28, 16>), grid, block, plan->fast_lds_bytes, stream, fp);
      else if (plan->nrows == 19) hipLaunchKernelGGL((fft2048c_kernel<19, false>), grid, block, plan->fast_lds_bytes, stream, fp);
      else hipLaunchKernelGGL((fft2048c_kernel<32, false>), grid, block, plan->fast_lds_bytes, stream, fp);
    }
    HIP_TRY(hipGetLastError());
    return HIPFEAT_OK;
  }
  if (plan->variant == 8) {
    Fft1024cParams fp{};
    fp.wave = d_wave;
    fp.out = d_out;
    fp.cuts = lay->d_cuts;
    fp.shared_consts = plan->d_c_shared;
    fp.out_stride = lay->out_row_stride;
    fp.num_cuts = (int32_t)lay->batch;
    fp.uniform_bpc = lay->uniform_bpc;
    fp.frames_per_block = lay->fpb;
    fp.rounds = lay->fpb / plan->fpb_unit;
    fp.N = c.frame_length;
    fp.shift = c.frame_shift;
    fp.npad_left = plan->npad_left;
    fp.M = c.num_filters;
    fp.flags = (c.remove_dc_offset ? F_REMOVE_DC : 0) | (c.use_fft_mag ? F_FFT_MAG : 0) | (c.kind == HIPFEAT_LIBROSA_FBANK ? (F_CENTER | F_LOG10) : 0);
    fp.preemph = c.preemph_coeff;
    fp.mel_floor = c.mel_floor;
    fp.shared_floats = plan->c_shared_floats;
    fp.wtab_off = plan->c_wtab_off;
    fp.ltab_off = plan->c_ltab_off;
    fp.xs_floats = plan->c_xs_floats;
    fp.nsets = plan->w_nsets;
    for (int s2 = 0; s2 < kWMaxSets; ++s2) fp.steps[s2] = plan->w_steps[s2], fp.step0[s2] = plan->w_step0[s2];
    DeviceGuard g(plan->device);
    const dim3 grid((unsigned)lay->total_blocks), block(64 * plan->w_waves);
    set_lds_poison(plan->fast_lds_bytes);
    if (plan->w_fixed == 1) hipLaunchKernelGGL((fft1024c_kernel<20, 24, 16, 8>), grid, block, plan->fast_lds_bytes, stream, fp);
    else if (plan->w_fixed == 2) hipLaunchKernelGGL((fft1024c_kernel<26, 24, 16, 8>), grid, block, plan->fast_lds_bytes, stream, fp);
    else if (plan->w_fixed == 3) hipLaunchKernelGGL((fft1024c_kernel<20, 24, 24, 8>), grid, block, plan->fast_lds_bytes, stream, fp);
    else if (plan->w_fixed == 4) hipLaunchKernelGGL((fft1024c_kernel<32, 16, 16, 8, true>), grid, block, plan->fast_lds_bytes, stream, fp);
    else if (plan->nrows == 20) hipLaunchKernelGGL(fft1024c_kernel<20>, grid, block, plan->fast_lds_bytes, stream, fp);
    else if (plan->nrows == 26) hipLaunchKernelGGL(fft1024c_kernel<26>, grid, block, plan->fast_lds_bytes, stream, fp);
    else hipLaunchKernelGGL(fft1024c_kernel<32>, grid, block, plan->fast_lds_bytes, stream, fp);
    HIP_TRY(hipGetLastError());
    return HIPFEAT_OK;
  }
  if (plan->variant == 7) {
    Fft512cParams fp{};
    fp.wave = d_wave;
    fp.out = d_out;
    fp.cuts = lay->d_cuts;
    fp.shared_consts = plan->d_c_shared;
    fp.out_stride = lay->out_row_stride;
    fp.num_cuts = (int32_t)lay->batch;
    fp.uniform_bpc = lay->uniform_bpc;
    fp.frames_per_block = lay->fpb;
    fp.rounds = lay->fpb / plan->fpb_unit;
    fp.N = c.frame_length;
    fp.shift = c.frame_shift;
    fp.npad_left = plan->npad_left;
    fp.M = c.num_filters;
    fp.flags = c.remove_dc_offset ? F_REMOVE_DC : 0;
    fp.preemph = c.preemph_coeff;
    fp.mel_floor = c.mel_floor;
    fp.shared_floats = plan->c_shared_floats;
    fp.wtab_off = plan->c_wtab_off;
    fp.ltab_off = plan->c_ltab_off;
    fp.xs_floats = plan->c_xs_floats;
    fp.dct_tab = plan->d_dct_consts;
    fp.C = c.num_ceps;
    fp.total_quads = (int32_t)lay->total_quads;
    DeviceGuard g(plan->device);
    const dim3 grid((unsigned)lay->total_blocks), block(64 * kCWaves);
    set_lds_poison(plan->fast_lds_bytes);
    fft512c_dispatch(plan->c_mode, plan->nrows, c.frame_length, true, grid, block, plan->fast_lds_bytes, stream, &fp, lay->flat);
    HIP_TRY(hipGetLastError());
    return HIPFEAT_OK;
  }
  if (plan->variant == 2 || plan->variant == 4) {
    Fft512Params fp{};
    fp.wave = d_wave;
    fp.out = d_out;
    fp.cuts = lay->d_cuts;
    fp.lds_consts = plan->d_lds_consts;
    fp.mel_a = plan->d_mel_a4;
    fp.work = plan->d_work;
    fp.out_stride = lay->out_row_stride;
    fp.num_cuts = (int32_t)lay->batch;
    fp.uniform_bpc = lay->uniform_bpc;
    fp.tiles_per_block = plan->tiles_per_block;
    fp.N = c.frame_length;
    fp.shift = c.frame_shift;
    fp.npad_left = plan->npad_left;
    fp.M = c.num_filters;
    fp.flags = (c.remove_dc_offset ? F_REMOVE_DC : 0) | (c.use_fft_mag ? F_FFT_MAG : 0) | (c.kind == HIPFEAT_LOG_SPECTROGRAM ? F_LOG_SPEC : 0);
    fp.log_offset = c.log_offset;
    fp.preemph = c.preemph_coeff;
    fp.mel_floor = c.mel_floor;
    fp.xs_floats = plan->xs_floats;
    fp.const_floats = plan->const_floats;
    fp.dct_consts = plan->d_dct_consts;
    fp.C = c.num_ceps;
    fp.lm_stride = plan->lm_stride;
    fp.dct_groups = plan->dct_groups;
    fp.dct_floats = plan->dct_floats;
    DeviceGuard g(plan->device);
    const dim3 grid((unsigned)lay->total_blocks), block(256);
    set_lds_poison(plan->fast_lds_bytes);
    if (plan->variant == 4) {
#define HF_LAUNCH_256(NR, OUT) hipLaunchKernelGGL((fft256_kernel<NR, OUT>), grid, block, plan->fast_lds_bytes, stream, fp)
      if (plan->nrows == 13) {
        if (plan->fast_out == 1) HF_LAUNCH_256(13, 1);
        else if (plan->fast_out == 2) HF_LAUNCH_256(13, 2);
        else HF_LAUNCH_256(13, 0);
      } else {
        if (plan->fast_out == 1) HF_LAUNCH_256(16, 1);
        else if (plan->fast_out == 2) HF_LAUNCH_256(16, 2);
        else HF_LAUNCH_256(16, 0);
      }
#undef HF_LAUNCH_256
    } else if (plan->variant == 2) {
#define HF_LAUNCH_B(NR, OUT) hipLaunchKernelGGL((fft512b_kernel<NR, OUT>), grid, block, plan->fast_lds_bytes, stream, fp)
#define HF_LAUNCH_NR(OUT)                      \
  do {                                         \
    if (plan->nrows == 10) HF_LAUNCH_B(10, OUT);      \
    else if (plan->nrows == 13) HF_LAUNCH_B(13, OUT); \
    else HF_LAUNCH_B(16, OUT);                        \
  } while (0)
      if (plan->fast_out == 1) HF_LAUNCH_NR(1);
      else if (plan->fast_out == 2) HF_LAUNCH_NR(2);
      else HF_LAUNCH_NR(0);
#undef HF_LAUNCH_NR
#undef HF_LAUNCH_B
    }
    HIP_TRY(hipGetLastError());
    return HIPFEAT_OK;
  }
  GenericParams gp{};
  gp.wave = d_wave;
  gp.out = d_out;
  gp.cuts = lay->d_cuts;
  gp.window = plan->d_window;
  gp.tw = plan->d_tw;
  gp.mel = plan->d_mel;
  gp.mel_range = plan->d_mel_range;
  gp.dct = plan->d_dct;
  gp.lifter = plan->d_lifter;
  gp.out_stride = lay->out_row_stride;
  gp.num_cuts = (int32_t)lay->batch;
  gp.uniform_bpc = lay->uniform_bpc;
  gp.N = c.frame_length;
  gp.shift = c.frame_shift;
  gp.fft = c.fft_length;
  gp.H = plan->H;
  gp.log2H = plan->log2H;
  gp.K = plan->K;
  gp.M = c.num_filters;
  gp.C = c.num_ceps;
  const bool whisper = c.kind == HIPFEAT_WHISPER, librosa = c.kind == HIPFEAT_LIBROSA_FBANK;
  gp.kind = (whisper || librosa) ? (int)HIPFEAT_FBANK : c.kind;  // same epilogue with log10; the post-pass below finishes it
  gp.flags = (c.remove_dc_offset ? F_REMOVE_DC : 0) | (c.use_energy ? F_USE_ENERGY : 0) | (c.raw_energy ? F_RAW_ENERGY : 0) |
             (c.use_fft_mag ? F_FFT_MAG : 0) | (c.apply_lifter ? F_LIFTER : 0) | (plan->pow2 ? F_POW2 : 0) | ((whisper || librosa) ? (F_CENTER | F_LOG10) : 0);
  gp.fpb = plan->fpb;
  gp.npad_left = plan->npad_left;
  gp.preemph = c.preemph_coeff;
  gp.log_energy_floor = c.energy_floor > 0.0f ? logf(c.energy_floor) : -INFINITY;
  gp.mel_floor = c.mel_floor;
  gp.log_offset = c.log_offset;
  gp.span = plan->span;
  gp.off_z = plan->off_z;
  gp.off_p = plan->off_p;
  gp.off_tw = plan->off_tw;
  gp.off_stat = plan->off_stat;
  gp.off_mel = plan->off_mel;
  DeviceGuard g(plan->device);
  set_lds_poison(plan->lds_bytes);
  hipLaunchKernelGGL(generic_kernel, dim3((unsigned)lay->total_blocks), dim3(256), plan->lds_bytes, stream, gp);
  HIP_TRY(hipGetLastError());
  if (whisper) {
    hipLaunchKernelGGL(whisper_norm_kernel, dim3((unsigned)lay->batch), dim3(1024), 0, stream, lay->d_cuts, d_out, lay->out_row_stride,
                       (int32_t)c.num_filters, (int32_t)c.frame_shift);
    HIP_TRY(hipGetLastError());
  }
  return HIPFEAT_OK;
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_extract_layout(const hipfeat_plan* plan, const hipfeat_layout* layout,
                                                 const float* d_wave, float* d_out, void* stream) {
  if (!plan || !layout) return fail(HIPFEAT_ERR_INVALID, "plan/layout is NULL");
  return launch(plan, layout, d_wave, d_out, (hipStream_t)stream);
}

// fills rows [num_frames, rows_per_cut) of every cut's slot in a collated (B, rows_per_cut, F) output
__global__ __launch_bounds__(256) void fill_padding_kernel(const CutDesc* __restrict__ cuts, float* __restrict__ out, int64_t row_stride,
                                                           int32_t rows_per_cut, int32_t feature_dim, float value) {
  const CutDesc cd = cuts[blockIdx.y];
  const int64_t n = (int64_t)(rows_per_cut - cd.num_frames) * feature_dim;
  float* __restrict__ base = out + (cd.out_row + cd.num_frames) * row_stride;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / feature_dim;
    base[r * row_stride + (i - r * feature_dim)] = value;
  }
}

__global__ __launch_bounds__(256) void pcm16_to_float_kernel(const int16_t* __restrict__ in, float* __restrict__ out, int64_t n) {
  // 8 samples per lane per step: one 16-byte load, two 16-byte stores
  const int64_t n8 = n >> 3;
  const bool aligned = ((reinterpret_cast<uintptr_t>(in) & 15) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  constexpr float k = 1.0f / 32768.0f;
  if (aligned) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
      const int4 v = reinterpret_cast<const int4*>(in)[i];
      const int w[4] = {v.x, v.y, v.z, v.w};
      float4 a, b;
      a.x = (float)(int16_t)(w[0] & 0xffff) * k; a.y = (float)(w[0] >> 16) * k;
      a.z = (float)(int16_t)(w[1] & 0xffff) * k; a.w = (float)(w[1] >> 16) * k;
      b.x = (float)(int16_t)(w[2] & 0xffff) * k; b.y = (float)(w[2] >> 16) * k;
      b.z = (float)(int16_t)(w[3] & 0xffff) * k; b.w = (float)(w[3] >> 16) * k;
      reinterpret_cast<float4*>(out)[2 * i] = a;
      reinterpret_cast<float4*>(out)[2 * i + 1] = b;
    }
    for (int64_t i = (n8 << 3) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = (float)in[i] * k;
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = (float)in[i] * k;
  }
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_pcm16_to_float(const int16_t* d_pcm, float* d_wave, int64_t num_samples, void* stream) {
  if (num_samples < 0 || (num_samples > 0 && (!d_pcm || !d_wave))) return fail(HIPFEAT_ERR_INVALID, "bad pcm16 arguments");
  if (num_samples == 0) return HIPFEAT_OK;
  const int64_t blocks = std::min<int64_t>((num_samples / 8 + 255) / 256 + 1, 256 * 32);
  hipLaunchKernelGGL(pcm16_to_float_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_pcm, d_wave, num_samples);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(HIPFEAT_ERR_HIP, "pcm16 launch failed: %s", hipGetErrorName(e));
  return HIPFEAT_OK;
}

__global__ __launch_bounds__(256) void float_to_half_kernel(const float* __restrict__ in, uint16_t* __restrict__ out, int64_t n) {
  // 8 values per lane per step (two 16-byte loads, one 16-byte store) when both buffers are 16-byte aligned; v_cvt_pkrtz would round
  // towards zero, so the conversion is the scalar round-to-nearest-even one
  const int64_t n8 = n >> 3;
  const bool aligned = ((reinterpret_cast<uintptr_t>(in) & 15) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  auto cvt = [](float x) -> unsigned { return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)x); };
  if (aligned) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
      const float4 a = reinterpret_cast<const float4*>(in)[2 * i], b = reinterpret_cast<const float4*>(in)[2 * i + 1];
      uint4 o;
      o.x = cvt(a.x) | (cvt(a.y) << 16);
      o.y = cvt(a.z) | (cvt(a.w) << 16);
      o.z = cvt(b.x) | (cvt(b.y) << 16);
      o.w = cvt(b.z) | (cvt(b.w) << 16);
      reinterpret_cast<uint4*>(out)[i] = o;
    }
    for (int64_t i = (n8 << 3) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = (uint16_t)cvt(in[i]);
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = (uint16_t)cvt(in[i]);
  }
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_float_to_half(const float* d_in, uint16_t* d_out, int64_t n, void* stream) {
  if (n < 0 || (n > 0 && (!d_in || !d_out))) return fail(HIPFEAT_ERR_INVALID, "bad float_to_half arguments");
  if (n == 0) return HIPFEAT_OK;
  const int64_t blocks = std::min<int64_t>((n / 8 + 255) / 256 + 1, 256 * 32);
  hipLaunchKernelGGL(float_to_half_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_in, d_out, n);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(HIPFEAT_ERR_HIP, "float_to_half launch failed: %s", hipGetErrorName(e));
  return HIPFEAT_OK;
}

// transient layout: descriptors staged through a pinned ring slot so the call stays asynchronous
static hipfeat_status extract_transient(const hipfeat_plan* plan, const float* d_wave, const int64_t* h_wave_offsets,
                                        const int64_t* h_num_samples, const int64_t* h_padded_len, int64_t batch, float* d_out,
                                        const int64_t* h_out_rows, int64_t out_row_stride, void* stream, int64_t rows_per_cut,
                                        float pad_value, int64_t* h_num_frames) {
  if (!plan) return fail(HIPFEAT_ERR_INVALID, "plan is NULL");
  hipfeat_layout lay;
  lay.owns = false;
  std::vector<CutDesc> descs;
  hipfeat_status st = build_descs(plan, batch, h_wave_offsets, h_num_samples, h_padded_len, h_out_rows, out_row_stride, descs, &lay);
  if (st != HIPFEAT_OK) return st;
  int64_t max_pad = 0;
  if (rows_per_cut >= 0) {
    for (int64_t b = 0; b < batch; ++b) {
      if (lay.num_frames[(size_t)b] > rows_per_cut)
        return fail(HIPFEAT_ERR_INVALID, "cut %lld has %lld frames but the collated output holds %lld rows per cut", (long long)b,
                    (long long)lay.num_frames[(size_t)b], (long long)rows_per_cut);
      max_pad = std::max(max_pad, rows_per_cut - lay.num_frames[(size_t)b]);
    }
  }
  if (h_num_frames)
    for (int64_t b = 0; b < batch; ++b) h_num_frames[b] = lay.num_frames[(size_t)b];
  if (lay.total_blocks == 0 && max_pad == 0) return HIPFEAT_OK;
  DeviceGuard g(plan->device);
  // Whisper with the fused normalisation: one armed scratch entry per cut travels behind the descriptors (fresh for every call)
  const size_t map_bytes = (lay.block_cut.size() * sizeof(int32_t) + 15) & ~(size_t)15;  // ragged: workgroup -> cut map behind the descriptors
  const size_t desc_bytes = descs.size() * sizeof(CutDesc) + map_bytes;
  const size_t bytes = desc_bytes + (plan->variant == 9 ? norm_slot_bytes(&lay) : 0);
  std::lock_guard<std::mutex> lk(plan->mu);
  StagingSlot& s = plan->slots[plan->next_slot];
  plan->next_slot = (plan->next_slot + 1) % 4;
  if (s.busy) {
    HIP_TRY(hipEventSynchronize(s.ev));
    s.busy = false;
  }
  if (!s.ev) HIP_TRY(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
  if (s.cap < bytes) {
    if (s.h) (void)hipHostFree(s.h);
    if (s.d) (void)hipFree(s.d);
    s.h = s.d = nullptr;
    s.cap = 0;
    const size_t cap = std::max<size_t>(bytes * 2, 1 << 16);
    HIP_TRY(hipHostMalloc(&s.h, cap, hipHostMallocDefault));
    HIP_TRY(hipMalloc(&s.d, cap));
    s.cap = cap;
  }
  std::memcpy(s.h, descs.data(), descs.size() * sizeof(CutDesc));
  if (map_bytes) std::memcpy(static_cast<unsigned char*>(s.h) + descs.size() * sizeof(CutDesc), lay.block_cut.data(), lay.block_cut.size() * sizeof(int32_t));
  if (bytes > desc_bytes) {
    std::memset(static_cast<unsigned char*>(s.h) + desc_bytes, 0, bytes - desc_bytes);
    lay.d_norm = static_cast<unsigned char*>(s.d) + desc_bytes;
    lay.norm_slots = 1;
  }
  HIP_TRY(hipMemcpyAsync(s.d, s.h, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  lay.d_cuts = static_cast<CutDesc*>(s.d);
  st = HIPFEAT_OK;
  if (lay.total_blocks > 0) st = launch(plan, &lay, d_wave, d_out, (hipStream_t)stream);
  if (st == HIPFEAT_OK && max_pad > 0) {
    const int64_t per_cut = max_pad * plan->feature_dim;
    const unsigned gx = (unsigned)std::min<int64_t>((per_cut + 1023) / 1024, 64);
    hipLaunchKernelGGL(fill_padding_kernel, dim3(gx, (unsigned)batch), dim3(256), 0, (hipStream_t)stream, lay.d_cuts, d_out,
                       out_row_stride, (int32_t)rows_per_cut, (int32_t)plan->feature_dim, pad_value);
    hipError_t e1 = hipGetLastError();
    if (e1 != hipSuccess) st = fail(HIPFEAT_ERR_HIP, "padding fill launch failed: %s", hipGetErrorName(e1));
  }
  hipError_t e = hipEventRecord(s.ev, (hipStream_t)stream);
  s.busy = (e == hipSuccess);
  return st;
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_extract(const hipfeat_plan* plan, const float* d_wave, const int64_t* h_wave_offsets,
                                          const int64_t* h_num_samples, const int64_t* h_padded_len, int64_t batch,
                                          float* d_out, const int64_t* h_out_rows, int64_t out_row_stride, void* stream) {
  return extract_transient(plan, d_wave, h_wave_offsets, h_num_samples, h_padded_len, batch, d_out, h_out_rows, out_row_stride, stream, -1,
                           0.f, nullptr);
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_extract_collated(const hipfeat_plan* plan, const float* d_wave, const int64_t* h_wave_offsets,
                                                   const int64_t* h_num_samples, const int64_t* h_padded_len, int64_t batch,
                                                   float* d_out, int64_t rows_per_cut, float pad_value, int64_t* h_num_frames,
                                                   void* stream) {
  if (!plan) return fail(HIPFEAT_ERR_INVALID, "plan is NULL");
  if (rows_per_cut < 0 || batch < 0 || batch > 65535) return fail(HIPFEAT_ERR_INVALID, "collated output: bad rows_per_cut / batch (max 65535 cuts)");
  std::vector<int64_t> rows((size_t)batch);
  for (int64_t b = 0; b < batch; ++b) rows[(size_t)b] = b * rows_per_cut;
  return extract_transient(plan, d_wave, h_wave_offsets, h_num_samples, h_padded_len, batch, d_out, rows.data(), plan->feature_dim, stream,
                           rows_per_cut, pad_value, h_num_frames);
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_extract_host(const hipfeat_plan* plan, const float* h_wave, int64_t wave_elems,
                                               const int64_t* h_wave_offsets, const int64_t* h_num_samples,
                                               const int64_t* h_padded_len, int64_t batch, float* h_out,
                                               int64_t out_elems, const int64_t* h_out_rows, int64_t out_row_stride,
                                               void* stream) {
  if (!plan) return fail(HIPFEAT_ERR_INVALID, "plan is NULL");
  if (wave_elems < 0 || out_elems < 0 || (wave_elems && !h_wave) || (out_elems && !h_out))
    return fail(HIPFEAT_ERR_INVALID, "bad host buffers");
  for (int64_t b = 0; b < batch; ++b) {
    if (h_wave_offsets[b] < 0 || h_wave_offsets[b] + h_num_samples[b] > wave_elems)
      return fail(HIPFEAT_ERR_INVALID, "cut %lld lies outside the waveform buffer", (long long)b);
  }
  for (int64_t b = 0, row = 0; b < batch; ++b) {
    const hipfeat_config& c = plan->cfg;
    int64_t T = hipfeat_num_frames(h_num_samples[b], c.frame_length, c.frame_shift, c.snip_edges);
    if (h_padded_len) {
      const int64_t hop = c.batch_hop > 0 ? c.batch_hop : c.frame_shift;
      T = std::min<int64_t>((h_num_samples[b] + hop / 2) / hop, hipfeat_num_frames(h_padded_len[b], c.frame_length, c.frame_shift, c.snip_edges));
    }
    const int64_t r0 = h_out_rows ? h_out_rows[b] : row;
    if (r0 < 0 || (T > 0 && ((r0 + T - 1) * out_row_stride + plan->feature_dim) > out_elems))
      return fail(HIPFEAT_ERR_INVALID, "cut %lld: output rows lie outside the output buffer", (long long)b);
    row += T;
  }
  DeviceGuard g(plan->device);
  hipStream_t st_ = (hipStream_t)stream;
  float *dw, *dout;
  {
    std::lock_guard<std::mutex> lk(plan->mu);
    if (plan->scratch_wave_cap < (size_t)wave_elems) {
      (void)hipFree(plan->d_scratch_wave);
      plan->d_scratch_wave = nullptr;
      plan->scratch_wave_cap = 0;
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&plan->d_scratch_wave), std::max<size_t>(wave_elems, 1) * sizeof(float)));
      plan->scratch_wave_cap = (size_t)wave_elems;
    }
    if (plan->scratch_out_cap < (size_t)out_elems) {
      (void)hipFree(plan->d_scratch_out);
      plan->d_scratch_out = nullptr;
      plan->scratch_out_cap = 0;
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&plan->d_scratch_out), std::max<size_t>(out_elems, 1) * sizeof(float)));
      plan->scratch_out_cap = (size_t)out_elems;
    }
    dw = plan->d_scratch_wave;
    dout = plan->d_scratch_out;
  }
  // rows / columns no cut covers come back as zeros
  if (out_elems) HIP_TRY(hipMemsetAsync(dout, 0, (size_t)out_elems * sizeof(float), st_));
  if (wave_elems) HIP_TRY(hipMemcpyAsync(dw, h_wave, (size_t)wave_elems * sizeof(float), hipMemcpyHostToDevice, st_));
  hipfeat_status st = hipfeat_extract(plan, dw, h_wave_offsets, h_num_samples, h_padded_len, batch, dout, h_out_rows,
                                      out_row_stride, stream);
  if (st != HIPFEAT_OK) return st;
  if (out_elems) HIP_TRY(hipMemcpyAsync(h_out, dout, (size_t)out_elems * sizeof(float), hipMemcpyDeviceToHost, st_));
  HIP_TRY(hipStreamSynchronize(st_));
  return HIPFEAT_OK;
}

// --------------------------------------------------------------------------------------
// post-feature transforms on the collated batch (GlobalMVN, SpecAugment)
// --------------------------------------------------------------------------------------
extern "C" HIPFEAT_API hipfeat_status hipfeat_global_mvn(const float* d_in, float* d_out, const float* d_means, const float* d_stds,
                                                         int64_t rows, int64_t feature_dim, int inverse, void* stream) {
  if (rows < 0 || feature_dim <= 0 || feature_dim > INT32_MAX || (rows > 0 && (!d_in || !d_out)) || !d_means || !d_stds)
    return fail(HIPFEAT_ERR_INVALID, "bad global_mvn arguments");
  const int64_t n = rows * feature_dim;
  if (n == 0) return HIPFEAT_OK;
  const int64_t blocks = std::min<int64_t>((n + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(global_mvn_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_in, d_out, d_means, d_stds, n,
                     (int32_t)feature_dim, (int32_t)(inverse != 0));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(HIPFEAT_ERR_HIP, "global_mvn launch failed: %s", hipGetErrorName(e));
  return HIPFEAT_OK;
}

namespace {
struct StagingRing {
  std::mutex mu;
  StagingSlot slots[4];
  int next = 0;
};
StagingRing g_rings[64];  // per device, process lifetime (descriptors of plan-less calls)
}  // namespace

extern "C" HIPFEAT_API hipfeat_status hipfeat_specaug(const float* d_in, float* d_out, int64_t batch, int64_t num_frames, int64_t feature_dim,
                                                      const hipfeat_warp_segment* h_segments, int64_t num_segments,
                                                      const hipfeat_mask* h_masks, int64_t num_masks, void* stream) {
  if (batch < 0 || num_frames < 0 || feature_dim < 0 || num_segments < 0 || num_masks < 0 || (num_segments && !h_segments) ||
      (num_masks && !h_masks))
    return fail(HIPFEAT_ERR_INVALID, "bad specaug arguments");
  if (batch > 65535 || num_frames > (1 << 24) || feature_dim > 32768)
    return fail(HIPFEAT_ERR_INVALID, "specaug: batch <= 65535, num_frames <= 2^24, feature_dim <= 32768");
  if (batch * num_frames * feature_dim == 0) return HIPFEAT_OK;
  if (!d_in || !d_out || d_in == d_out) return fail(HIPFEAT_ERR_INVALID, "specaug: needs distinct input and output buffers");
  const int B = (int)batch, T = (int)num_frames, F = (int)feature_dim;
  // group by sequence (stable), validate
  std::vector<int32_t> seg_off((size_t)B + 1, 0), mask_off((size_t)B + 1, 0);
  for (int64_t i = 0; i < num_segments; ++i) {
    const hipfeat_warp_segment& g = h_segments[i];
    if (g.sequence < 0 || g.sequence >= B || g.start < 0 || g.num_frames < 2 || g.start + (int64_t)g.num_frames > T || g.center < 1 ||
        g.center >= g.num_frames || g.warped < 1 || g.warped >= g.num_frames)
      return fail(HIPFEAT_ERR_INVALID, "specaug: warp segment %lld is out of range", (long long)i);
    seg_off[(size_t)g.sequence + 1]++;
  }
  for (int64_t i = 0; i < num_masks; ++i) {
    const hipfeat_mask& m = h_masks[i];
    if (m.sequence < 0 || m.sequence >= B || (m.axis != 1 && m.axis != 2) || m.begin < 0 || m.end < m.begin)
      return fail(HIPFEAT_ERR_INVALID, "specaug: mask %lld is out of range", (long long)i);
    mask_off[(size_t)m.sequence + 1]++;
  }
  for (int b = 0; b < B; ++b) {
    seg_off[(size_t)b + 1] += seg_off[(size_t)b];
    mask_off[(size_t)b + 1] += mask_off[(size_t)b];
  }
  std::vector<WarpSeg> segs((size_t)num_segments);
  std::vector<int32_t> masks((size_t)num_masks * 3);
  {
    std::vector<int32_t> cur(seg_off.begin(), seg_off.end() - 1);
    for (int64_t i = 0; i < num_segments; ++i) {
      const hipfeat_warp_segment& g = h_segments[i];
      segs[(size_t)cur[(size_t)g.sequence]++] = WarpSeg{g.sequence, g.start, g.num_frames, g.center, g.warped};
    }
    for (int b = 0; b < B; ++b)  // the reference warps segment after segment: overlapping ones would read each other's output
      for (int i = seg_off[(size_t)b]; i < seg_off[(size_t)b + 1]; ++i)
        for (int j = i + 1; j < seg_off[(size_t)b + 1]; ++j)
          if (segs[(size_t)i].start < segs[(size_t)j].start + segs[(size_t)j].len && segs[(size_t)j].start < segs[(size_t)i].start + segs[(size_t)i].len)
            return fail(HIPFEAT_ERR_INVALID, "specaug: warp segments of sequence %d overlap (apply them in separate calls)", b);
    std::vector<int32_t> mc(mask_off.begin(), mask_off.end() - 1);
    for (int64_t i = 0; i < num_masks; ++i) {
      const hipfeat_mask& m = h_masks[i];
      const size_t k = (size_t)mc[(size_t)m.sequence]++ * 3;
      masks[k] = m.axis;
      masks[k + 1] = m.begin;
      masks[k + 2] = m.end;
    }
  }
  int tile_elems = 8192;
  const int rows_per_tile = std::max(1, std::min(T, tile_elems / std::max(F, 1)));
  const int tiles = (T + rows_per_tile - 1) / rows_per_tile;
  auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
  const size_t o_seg_off = 0, o_mask_off = up16(o_seg_off + seg_off.size() * 4), o_segs = up16(o_mask_off + mask_off.size() * 4),
               o_masks = up16(o_segs + segs.size() * sizeof(WarpSeg)), o_part = up16(o_masks + masks.size() * 4),
               h_bytes = o_part, total = o_part + (size_t)B * tiles * 4;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return fail(HIPFEAT_ERR_INVALID, "device index %d out of range", dev);
  StagingRing& ring = g_rings[dev];
  std::lock_guard<std::mutex> lk(ring.mu);
  StagingSlot& sl = ring.slots[ring.next];
  ring.next = (ring.next + 1) % 4;
  if (sl.busy) {
    HIP_TRY(hipEventSynchronize(sl.ev));
    sl.busy = false;
  }
  if (!sl.ev) HIP_TRY(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
  if (sl.cap < total) {
    if (sl.h) (void)hipHostFree(sl.h);
    if (sl.d) (void)hipFree(sl.d);
    sl.h = sl.d = nullptr;
    sl.cap = 0;
    const size_t cap = std::max<size_t>(total * 2, 1 << 16);
    HIP_TRY(hipHostMalloc(&sl.h, cap, hipHostMallocDefault));
    HIP_TRY(hipMalloc(&sl.d, cap));
    sl.cap = cap;
  }
  char* h = static_cast<char*>(sl.h);
  std::memcpy(h + o_seg_off, seg_off.data(), seg_off.size() * 4);
  std::memcpy(h + o_mask_off, mask_off.data(), mask_off.size() * 4);
  if (!segs.empty()) std::memcpy(h + o_segs, segs.data(), segs.size() * sizeof(WarpSeg));
  if (!masks.empty()) std::memcpy(h + o_masks, masks.data(), masks.size() * 4);
  hipStream_t st_ = (hipStream_t)stream;
  HIP_TRY(hipMemcpyAsync(sl.d, sl.h, h_bytes, hipMemcpyHostToDevice, st_));
  char* d = static_cast<char*>(sl.d);
  SpecAugParams sp{};
  sp.in = d_in;
  sp.out = d_out;
  sp.B = B;
  sp.T = T;
  sp.F = F;
  sp.seg_off = reinterpret_cast<const int32_t*>(d + o_seg_off);
  sp.mask_off = reinterpret_cast<const int32_t*>(d + o_mask_off);
  sp.segs = reinterpret_cast<const WarpSeg*>(d + o_segs);
  sp.masks = reinterpret_cast<const int32_t*>(d + o_masks);
  sp.partials = reinterpret_cast<float*>(d + o_part);
  sp.tiles = tiles;
  sp.rows_per_tile = rows_per_tile;
  const bool vec4 = F % 4 == 0 && ((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_out)) & 15) == 0;
  if (vec4) hipLaunchKernelGGL(specaug_warp_kernel<4>, dim3((unsigned)tiles, (unsigned)B), dim3(256), 0, st_, sp);
  else hipLaunchKernelGGL(specaug_warp_kernel<1>, dim3((unsigned)tiles, (unsigned)B), dim3(256), 0, st_, sp);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && num_masks > 0) {
    hipLaunchKernelGGL(specaug_mask_kernel, dim3((unsigned)tiles, (unsigned)B), dim3(256), (size_t)F + rows_per_tile, st_, sp);
    e = hipGetLastError();
  }
  hipError_t e2 = hipEventRecord(sl.ev, st_);
  sl.busy = (e2 == hipSuccess);
  if (e != hipSuccess) return fail(HIPFEAT_ERR_HIP, "specaug launch failed: %s", hipGetErrorName(e));
  return HIPFEAT_OK;
}

// --------------------------------------------------------------------------------------
// resampler (speed perturbation)
// --------------------------------------------------------------------------------------
struct hipfeat_resampler {
  int device = 0, orig = 0, nw = 0, kw = 0, width = 0;
  float* d_kernel = nullptr;
  int outs_per_block = 2048, span_floats = 0, kernel_in_lds = 0;
  size_t lds_bytes = 0;
  // fast path (compile-time ratio): transposed bank [kw][newp] + launcher
  float* d_kernel_t = nullptr;
  void (*fast)(const float*, float*, const ResCut*, const float*, int, unsigned, hipStream_t) = nullptr;
  int fast_outs = 0;
  mutable std::mutex mu;
  mutable StagingSlot slots[4];
  mutable int next_slot = 0;
};

template <int ORIG, int NEW, int WIDTH>
static void launch_resample_fast(const float* in, float* out, const ResCut* cuts, const float* kt, int num_cuts, unsigned blocks,
                                 hipStream_t stream) {
  hipLaunchKernelGGL((resample_fast_kernel<ORIG, NEW, WIDTH>), dim3(blocks), dim3(256), 0, stream, in, out, cuts, kt, num_cuts);
}

template <int ORIG, int NEW, int WIDTH>
static bool pick_resample_fast(hipfeat_resampler* r) {
  if (r->orig != ORIG || r->nw != NEW || r->width != WIDTH) return false;
  r->fast = &launch_resample_fast<ORIG, NEW, WIDTH>;
  r->fast_outs = ResampleFast<ORIG, NEW, WIDTH>::OUTS;
  return true;
}

extern "C" HIPFEAT_API int64_t hipfeat_resampled_length(int64_t num_samples, int32_t orig_freq, int32_t new_freq) {
  if (orig_freq <= 0 || new_freq <= 0 || num_samples < 0) return 0;
  // torch.ceil(torch.as_tensor(new * length / orig)): the quotient is a Python float stored as float32
  const float q = (float)((double)new_freq * (double)num_samples / (double)orig_freq);
  return (int64_t)std::ceil(q);
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_resampler_create(int32_t orig_freq, int32_t new_freq, int32_t width,
                                                               const float* h_kernel, int32_t device, hipfeat_resampler** out) {
  if (!out || !h_kernel) return fail(HIPFEAT_ERR_INVALID, "NULL argument");
  *out = nullptr;
  if (orig_freq <= 0 || new_freq <= 0 || width <= 0) return fail(HIPFEAT_ERR_INVALID, "bad resampler geometry");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
    return fail(HIPFEAT_ERR_HIP, "device %d not available (%d HIP devices visible)", device, ndev);
  hipfeat_resampler* r = new (std::nothrow) hipfeat_resampler();
  if (!r) return fail(HIPFEAT_ERR_INVALID, "out of host memory");
  r->device = device;
  r->orig = orig_freq;
  r->nw = new_freq;
  r->width = width;
  r->kw = 2 * width + orig_freq;
  DeviceGuard g(device);
  hipfeat_status st = upload(&r->d_kernel, h_kernel, (size_t)r->nw * r->kw);
  if (st != HIPFEAT_OK) {
    delete r;
    return st;
  }
  // speed 0.9 / 1.1 / 0.95 / 1.05 at any rate, and the 1:2, 2:1, 3:1 rate conversions
  const bool fast = !route_env("HIPFEAT_RESAMPLE_GENERIC") &&
                    (pick_resample_fast<9, 10, 7>(r) || pick_resample_fast<11, 10, 7>(r) || pick_resample_fast<19, 20, 7>(r) ||
                     pick_resample_fast<21, 20, 7>(r) || pick_resample_fast<1, 2, 7>(r) || pick_resample_fast<2, 1, 13>(r) ||
                     pick_resample_fast<3, 1, 19>(r));
  if (fast) {
    const int newp = (r->nw + 3) & ~3;
    std::vector<float> kt((size_t)r->kw * newp, 0.f);
    for (int ph = 0; ph < r->nw; ++ph)
      for (int i = 0; i < r->kw; ++i) kt[(size_t)i * newp + ph] = h_kernel[(size_t)ph * r->kw + i];
    st = upload(&r->d_kernel_t, kt.data(), kt.size());
    if (st != HIPFEAT_OK) {
      (void)hipFree(r->d_kernel);
      delete r;
      return st;
    }
  }
  r->kernel_in_lds = ((size_t)r->nw * r->kw <= 8192) ? 1 : 0;
  for (r->outs_per_block = 2048; r->outs_per_block >= 64; r->outs_per_block >>= 1) {
    r->span_floats = (((r->outs_per_block + r->nw - 1) / r->nw + 1) * r->orig + r->kw + 3) & ~3;
    r->lds_bytes = ((size_t)r->span_floats + (r->kernel_in_lds ? (size_t)r->nw * r->kw : 0)) * sizeof(float);
    if (r->lds_bytes <= 64 * 1024) break;
  }
  if (r->outs_per_block < 64) {
    (void)hipFree(r->d_kernel);
    delete r;
    return fail(HIPFEAT_ERR_UNSUPPORTED, "resampling ratio %d/%d needs too much LDS", orig_freq, new_freq);
  }
  *out = r;
  return HIPFEAT_OK;
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_resampler_destroy(hipfeat_resampler* r) {
  if (!r) return HIPFEAT_OK;
  DeviceGuard g(r->device);
  (void)hipFree(r->d_kernel);
  if (r->d_kernel_t) (void)hipFree(r->d_kernel_t);
  for (auto& s : r->slots) {
    if (s.h) (void)hipHostFree(s.h);
    if (s.d) (void)hipFree(s.d);
    if (s.ev) (void)hipEventDestroy(s.ev);
  }
  delete r;
  return HIPFEAT_OK;
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_resample(const hipfeat_resampler* r, const float* d_in, const int64_t* h_in_offsets,
                                                       const int64_t* h_num_samples, int64_t batch, float* d_out,
                                                       const int64_t* h_out_offsets, void* stream) {
  if (!r) return fail(HIPFEAT_ERR_INVALID, "resampler is NULL");
  if (batch < 0 || (batch > 0 && (!h_in_offsets || !h_num_samples || !h_out_offsets || !d_in || !d_out)))
    return fail(HIPFEAT_ERR_INVALID, "bad batch arguments");
  std::vector<ResCut> cuts((size_t)batch);
  int64_t blocks = 0;
  const int64_t opb = r->fast ? r->fast_outs : r->outs_per_block;
  for (int64_t b = 0; b < batch; ++b) {
    const int64_t L = h_num_samples[b];
    if (L < 0 || L > INT32_MAX / 2) return fail(HIPFEAT_ERR_INVALID, "cut %lld: %lld samples out of range", (long long)b, (long long)L);
    const int64_t ol = hipfeat_resampled_length(L, r->orig, r->nw);
    if (ol > INT32_MAX) return fail(HIPFEAT_ERR_INVALID, "cut %lld: output too long", (long long)b);
    cuts[(size_t)b] = ResCut{h_in_offsets[b], h_out_offsets[b], (int32_t)L, (int32_t)ol, (int32_t)blocks, 0};
    blocks += (ol + opb - 1) / opb;
    if (blocks > INT32_MAX - (1 << 24)) return fail(HIPFEAT_ERR_INVALID, "batch too large for one launch");
  }
  if (blocks == 0) return HIPFEAT_OK;
  DeviceGuard g(r->device);
  const size_t bytes = cuts.size() * sizeof(ResCut);
  std::lock_guard<std::mutex> lk(r->mu);
  StagingSlot& s = r->slots[r->next_slot];
  r->next_slot = (r->next_slot + 1) % 4;
  if (s.busy) {
    HIP_TRY(hipEventSynchronize(s.ev));
    s.busy = false;
  }
  if (!s.ev) HIP_TRY(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
  if (s.cap < bytes) {
    if (s.h) (void)hipHostFree(s.h);
    if (s.d) (void)hipFree(s.d);
    s.h = s.d = nullptr;
    s.cap = 0;
    const size_t cap = std::max<size_t>(bytes * 2, 1 << 16);
    HIP_TRY(hipHostMalloc(&s.h, cap, hipHostMallocDefault));
    HIP_TRY(hipMalloc(&s.d, cap));
    s.cap = cap;
  }
  std::memcpy(s.h, cuts.data(), bytes);
  HIP_TRY(hipMemcpyAsync(s.d, s.h, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  if (r->fast) {
    r->fast(d_in, d_out, static_cast<const ResCut*>(s.d), r->d_kernel_t, (int)batch, (unsigned)blocks, (hipStream_t)stream);
    hipError_t e1 = hipGetLastError();
    hipError_t e2 = hipEventRecord(s.ev, (hipStream_t)stream);
    s.busy = (e2 == hipSuccess);
    if (e1 != hipSuccess) return fail(HIPFEAT_ERR_HIP, "resample launch failed: %s", hipGetErrorName(e1));
    return HIPFEAT_OK;
  }
  ResampleParams rp{};
  rp.in = d_in;
  rp.out = d_out;
  rp.cuts = static_cast<const ResCut*>(s.d);
  rp.kernel = r->d_kernel;
  rp.num_cuts = (int32_t)batch;
  rp.orig = r->orig;
  rp.nw = r->nw;
  rp.kw = r->kw;
  rp.width = r->width;
  rp.outs_per_block = r->outs_per_block;
  rp.span_floats = r->span_floats;
  rp.kernel_in_lds = r->kernel_in_lds;
  hipLaunchKernelGGL(resample_kernel, dim3((unsigned)blocks), dim3(256), r->lds_bytes, (hipStream_t)stream, rp);
  hipError_t e1 = hipGetLastError();
  hipError_t e2 = hipEventRecord(s.ev, (hipStream_t)stream);
  s.busy = (e2 == hipSuccess);
  if (e1 != hipSuccess) return fail(HIPFEAT_ERR_HIP, "resample launch failed: %s", hipGetErrorName(e1));
  return HIPFEAT_OK;
}

// --------------------------------------------------------------------------------------
// on-the-fly mini-batch: mixed-factor speed perturbation + collated extraction in two launches (kernel_minibatch.hpp)
// --------------------------------------------------------------------------------------
constexpr int kMbSlots = 16;
constexpr int kMbMaxResamplers = 8;

struct MbSlot {
  int64_t ticket = -1;
  bool planned = false;
  const hipfeat_plan* plan = nullptr;
  hipfeat_layout lay;
  std::vector<CutDesc> descs;
  std::vector<ResCut> res;
  std::vector<int32_t> rows;        // rows per cut of the collated tensor each cut belongs to (grouped plans: fixed at plan time)
  std::vector<int64_t> group_first; // first cut of each group, then batch
  int64_t total_rows = 0;
  std::vector<int32_t> fill_first;  // prefix sum of the cuts' padding items (16 KB each)
  int64_t res_blocks = 0, arena_need = 0, max_frames = 0;  // res_blocks: resampler items (256 hops each) of the whole mini-batch
  void* h = nullptr;  // pinned staging (tables that do not fit the kernel arguments)
  void* d = nullptr;  // device tables: CutDesc[batch], then (staged path) ResCut[num_res]
  size_t cap = 0;
  hipEvent_t ev = nullptr;
  bool busy = false;
};

struct hipfeat_speed_bank {
  int device = 0;
  bool device_set = false;  // a bank without resamplers learns its device from the first plan it serves
  int num = 0;
  int kind[kMbMaxResamplers] = {};  // kernel_minibatch.hpp's switch index of resampler i
  int orig[kMbMaxResamplers] = {}, nw[kMbMaxResamplers] = {}, outs[kMbMaxResamplers] = {};
  const float* kt[kMbKinds] = {};
  size_t lds_bytes = 0;
  bool allow_inline = true;
  std::mutex mu;
  MbSlot slots[kMbSlots];
  int64_t next_ticket = 0;
};

extern "C" HIPFEAT_API hipfeat_status hipfeat_speed_bank_create(const hipfeat_resampler* const* resamplers, int32_t num,
                                                                hipfeat_speed_bank** out) {
  if (!out) return fail(HIPFEAT_ERR_INVALID, "bank pointer is NULL");
  *out = nullptr;
  if (num < 0 || num > kMbMaxResamplers || (num > 0 && !resamplers)) return fail(HIPFEAT_ERR_INVALID, "a bank holds 0 ... %d resamplers", kMbMaxResamplers);
  hipfeat_speed_bank* b = new (std::nothrow) hipfeat_speed_bank();
  if (!b) return fail(HIPFEAT_ERR_INVALID, "out of host memory");
  static const int ratios[kMbKinds][3] = {{9, 10, 7}, {11, 10, 7}};
  static const size_t lds_floats[kMbKinds] = {ResampleFast<9, 10, 7>::LDS_FLOATS, ResampleFast<11, 10, 7>::LDS_FLOATS};
  static const int outs[kMbKinds] = {ResampleFast<9, 10, 7>::OUTS, ResampleFast<11, 10, 7>::OUTS};
  b->lds_bytes = 16;
  for (int i = 0; i < num; ++i) {
    const hipfeat_resampler* r = resamplers[i];
    int k = -1;
    if (r && r->d_kernel_t)
      for (int j = 0; j < kMbKinds; ++j)
        if (r->orig == ratios[j][0] && r->nw == ratios[j][1] && r->width == ratios[j][2]) k = j;
    if (k < 0 || (i > 0 && r->device != b->device)) {
      const int o = r ? r->orig : 0, n = r ? r->nw : 0;
      delete b;
      return fail(HIPFEAT_ERR_UNSUPPORTED, "resampler %d (%d -> %d) is not one of the compile-time ratios of the mixed launch (9:10, 11:10 = speed 0.9 / 1.1), "
                  "or lives on another device: use hipfeat_resample per factor", i, o, n);
    }
    if (i == 0) {
      b->device = r->device;
      b->device_set = true;
    }
    b->kind[i] = k;
    b->orig[i] = r->orig;
    b->nw[i] = r->nw;
    b->outs[i] = outs[k];
    b->kt[k] = r->d_kernel_t;
    b->lds_bytes = std::max(b->lds_bytes, lds_floats[k] * sizeof(float));
  }
  b->num = num;
  b->allow_inline = route_env("HIPFEAT_MB_NO_INLINE") == nullptr;
  *out = b;
  return HIPFEAT_OK;
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_speed_bank_destroy(hipfeat_speed_bank* b) {
  if (!b) return HIPFEAT_OK;
  if (!b->device_set) {  // never served a plan: no slot was ever allocated, and there is no device to touch
    delete b;
    return HIPFEAT_OK;
  }
  DeviceGuard g(b->device);
  for (auto& s : b->slots) {
    if (s.busy && s.ev) (void)hipEventSynchronize(s.ev);
    if (s.h) (void)hipHostFree(s.h);
    if (s.d) (void)hipFree(s.d);
    if (s.ev) (void)hipEventDestroy(s.ev);
  }
  delete b;
  return HIPFEAT_OK;
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_minibatch_plan(hipfeat_speed_bank* bank, const hipfeat_plan* plan, int64_t batch,
                                                             const int64_t* h_offsets, const int64_t* h_num_samples,
                                                             const int32_t* h_bank_index, const int64_t* h_max_samples,
                                                             int64_t tail_start, int32_t zero_pad_batch, int64_t num_groups,
                                                             const int64_t* h_group_sizes, int64_t* h_out_offsets,
                                                             int64_t* h_out_num_samples, int64_t* h_num_frames, int64_t* h_group_rows,
                                                             int64_t* h_info) {
  if (!bank || !plan || !h_info) return fail(HIPFEAT_ERR_INVALID, "NULL argument");
  if (batch <= 0 || !h_offsets || !h_num_samples || batch > 65535) return fail(HIPFEAT_ERR_INVALID, "bad batch arguments (1 ... 65535 cuts)");
  if (plan->variant == 9 || plan->cfg.kind == HIPFEAT_WHISPER || plan->cfg.kind == HIPFEAT_LIBROSA_FBANK)
    return fail(HIPFEAT_ERR_UNSUPPORTED, "the mini-batch launch pair serves the Kaldi-style plans (spectrogram / fbank / mfcc)");
  if (num_groups < 0 || (num_groups > 0 && !h_group_sizes)) return fail(HIPFEAT_ERR_INVALID, "bad group arguments");
  std::lock_guard<std::mutex> lk(bank->mu);
  if (!bank->device_set) {  // a bank without resamplers (plain collated extraction): its slots live on the device of the plans it serves
    bank->device = plan->device;
    bank->device_set = true;
  }
  if (plan->device != bank->device) return fail(HIPFEAT_ERR_INVALID, "plan (device %d) and bank (device %d) live on different devices", plan->device, bank->device);
  const int64_t ticket = bank->next_ticket++;
  MbSlot& s = bank->slots[ticket % kMbSlots];
  s.planned = false;
  s.ticket = ticket;
  s.plan = plan;
  s.res.clear();
  s.group_first.assign(1, 0);
  if (num_groups > 0) {
    for (int64_t k = 0; k < num_groups; ++k) {
      if (h_group_sizes[k] <= 0) return fail(HIPFEAT_ERR_INVALID, "group %lld is empty", (long long)k);
      s.group_first.push_back(s.group_first.back() + h_group_sizes[k]);
    }
    if (s.group_first.back() != batch) return fail(HIPFEAT_ERR_INVALID, "the groups hold %lld cuts, the batch %lld", (long long)s.group_first.back(), (long long)batch);
  } else {
    s.group_first.push_back(batch);
  }
  // lengths and places of the perturbed batch: unperturbed cuts stay where they are, the others go to the tail in cut order, each on a
  // 16-byte boundary (the feature kernels fetch spans by LDS-DMA)
  std::vector<int64_t> offs((size_t)batch), lens((size_t)batch), padded;
  int64_t tail = (tail_start + 3) & ~(int64_t)3, blocks = 0, max_len = 0;
  for (int64_t b = 0; b < batch; ++b) {
    const int64_t L = h_num_samples[b];
    const int idx = h_bank_index ? h_bank_index[b] : -1;
    if (L < 0 || L > INT32_MAX / 2) return fail(HIPFEAT_ERR_INVALID, "cut %lld: %lld samples out of range", (long long)b, (long long)L);
    if (idx >= bank->num) return fail(HIPFEAT_ERR_INVALID, "cut %lld: bank index %d of %d", (long long)b, idx, bank->num);
    int64_t o = h_offsets[b], n = L;
    if (h_offsets[b] < 0 || h_offsets[b] + L > tail_start)  // (also the unperturbed cuts: the tail is about to be written)
      return fail(HIPFEAT_ERR_INVALID, "cut %lld (offset %lld, %lld samples) reaches into the arena's tail (tail_start %lld)", (long long)b, (long long)h_offsets[b],
                  (long long)L, (long long)tail_start);
    if (idx >= 0) {
      const int64_t ol = hipfeat_resampled_length(L, bank->orig[idx], bank->nw[idx]);
      s.res.push_back(ResCut{h_offsets[b], tail, (int32_t)L, (int32_t)ol, (int32_t)blocks, bank->kind[idx]});
      blocks += (ol + bank->outs[idx] - 1) / bank->outs[idx];
      if (blocks > INT32_MAX - (1 << 24)) return fail(HIPFEAT_ERR_INVALID, "batch too large for one launch");
      o = tail;
      n = ol;
      tail += (ol + 3) & ~(int64_t)3;
    }
    if (h_max_samples && h_max_samples[b] >= 0) n = std::min(n, h_max_samples[b]);  // a sample or two to truncate (lhotse/audio/recording.py:1058-1060)
    offs[(size_t)b] = o;
    lens[(size_t)b] = n;
    max_len = std::max(max_len, n);
  }
  if (zero_pad_batch) {  // edge_rule "batch_zero_pad" (_extract_batch, extractors.py:531-537): the longest cut of the cut's own mini-batch
    padded.resize((size_t)batch);
    for (size_t k = 0; k + 1 < s.group_first.size(); ++k) {
      int64_t m = 0;
      for (int64_t b = s.group_first[k]; b < s.group_first[k + 1]; ++b) m = std::max(m, lens[(size_t)b]);
      for (int64_t b = s.group_first[k]; b < s.group_first[k + 1]; ++b) padded[(size_t)b] = m;
    }
  }
  (void)max_len;
  hipfeat_status st = build_descs(plan, batch, offs.data(), lens.data(), zero_pad_batch ? padded.data() : nullptr, nullptr, plan->feature_dim, s.descs, &s.lay);
  if (st != HIPFEAT_OK) return st;
  s.lay.owns = false;
  s.res_blocks = blocks;
  s.arena_need = s.res.empty() ? tail_start : tail;  // (nothing to resample: the tail is not touched, its 16-byte alignment slack not needed)
  s.max_frames = 0;
  s.total_rows = 0;
  s.rows.resize((size_t)batch);
  for (size_t k = 0; k + 1 < s.group_first.size(); ++k) {  // every mini-batch = its own dense (B_k, T_k, F) tensor, back to back in d_out
    int64_t tk = 0;
    for (int64_t b = s.group_first[k]; b < s.group_first[k + 1]; ++b) tk = std::max(tk, s.lay.num_frames[(size_t)b]);
    for (int64_t b = s.group_first[k]; b < s.group_first[k + 1]; ++b) {
      s.rows[(size_t)b] = (int32_t)tk;
      s.descs[(size_t)b].out_row = s.total_rows + (b - s.group_first[k]) * tk;
    }
    if (h_group_rows) {
      h_group_rows[2 * k] = s.total_rows;
      h_group_rows[2 * k + 1] = tk;
    }
    s.total_rows += (s.group_first[k + 1] - s.group_first[k]) * tk;
    s.max_frames = std::max(s.max_frames, tk);
  }
  for (int64_t b = 0; b < batch; ++b) {
    if (h_out_offsets) h_out_offsets[b] = offs[(size_t)b];
    if (h_out_num_samples) h_out_num_samples[b] = lens[(size_t)b];
    if (h_num_frames) h_num_frames[b] = s.lay.num_frames[(size_t)b];
  }
  s.planned = true;
  h_info[0] = ticket;
  h_info[1] = s.arena_need;
  h_info[2] = s.max_frames;
  h_info[3] = s.total_rows;
  return HIPFEAT_OK;
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_minibatch_run(hipfeat_speed_bank* bank, int64_t ticket, float* d_arena, int64_t arena_floats,
                                                            float* d_out, int64_t rows_per_cut, float pad_value, void* stream) {
  if (!bank || !d_arena || !d_out) return fail(HIPFEAT_ERR_INVALID, "NULL argument");
  static const int dbg_skip = exp_env("HIPFEAT_MB_SKIP") ? atoi(exp_env("HIPFEAT_MB_SKIP")) : 0;  // experiments: 1 = no feature launch, 2 = no prep launch
  std::lock_guard<std::mutex> lk(bank->mu);
  MbSlot& s = bank->slots[((ticket % kMbSlots) + kMbSlots) % kMbSlots];
  if (s.ticket != ticket || !s.planned)
    return fail(HIPFEAT_ERR_INVALID, "ticket %lld is not a planned mini-batch (at most %d plans may be outstanding)", (long long)ticket, kMbSlots);
  s.planned = false;
  const hipfeat_plan* plan = s.plan;
  const int64_t batch = s.lay.batch;
  if (arena_floats < s.arena_need) return fail(HIPFEAT_ERR_INVALID, "arena holds %lld floats, the perturbed mini-batch needs %lld", (long long)arena_floats, (long long)s.arena_need);
  const bool grouped = s.group_first.size() > 2;
  if (grouped ? rows_per_cut >= 0 : rows_per_cut < s.max_frames)
    return fail(HIPFEAT_ERR_INVALID, grouped ? "a grouped plan fixes the rows per cut of every mini-batch (h_group_rows): pass rows_per_cut = -1 (got %lld; longest cut %lld frames)"
                                             : "the collated output holds %lld rows per cut, the longest cut has %lld frames", (long long)rows_per_cut, (long long)s.max_frames);
  if (!grouped)
    for (int64_t b = 0; b < batch; ++b) {
      s.descs[(size_t)b].out_row = b * rows_per_cut;
      s.rows[(size_t)b] = (int32_t)rows_per_cut;
    }
  DeviceGuard g(plan->device);
  hipStream_t st = (hipStream_t)stream;
  // padding items: 16 KB pieces of every cut's padding rows
  const int F = plan->feature_dim;
  s.fill_first.resize((size_t)batch + 1);
  int64_t fill_items = 0;
  for (int64_t b = 0; b < batch; ++b) {
    s.fill_first[(size_t)b] = (int32_t)fill_items;
    fill_items += ((int64_t)(s.rows[(size_t)b] - s.descs[(size_t)b].num_frames) * F + kMbFillFloats - 1) / kMbFillFloats;
    if (fill_items > INT32_MAX - (1 << 24)) return fail(HIPFEAT_ERR_INVALID, "batch too large for one launch");
  }
  s.fill_first[(size_t)batch] = (int32_t)fill_items;
  const size_t res_bytes = s.res.size() * sizeof(ResCut), cut_bytes = (size_t)batch * sizeof(CutDesc), fill_bytes = ((size_t)batch + 1) * sizeof(int32_t),
               row_bytes = (size_t)batch * sizeof(int32_t), bytes = (res_bytes + cut_bytes + fill_bytes + row_bytes + 15) & ~(size_t)15;
  const bool inl = bank->allow_inline && bytes <= (size_t)kMbInlineBytes;
  if (s.busy) {  // the launches that used this slot's device table last time
    HIP_TRY(hipEventSynchronize(s.ev));
    s.busy = false;
  }
  if (!s.ev) HIP_TRY(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
  const size_t map_bytes = ((size_t)s.lay.total_blocks * sizeof(int32_t) + 15) & ~(size_t)15;  // the feature launch's workgroup -> cut map
  const size_t dev_bytes = cut_bytes + map_bytes + bytes;  // the feature launch's CutDesc table and map, then (staged path) the blob
  if (s.cap < dev_bytes) {
    if (s.h) (void)hipHostFree(s.h);
    if (s.d) (void)hipFree(s.d);
    s.h = s.d = nullptr;
    s.cap = 0;
    const size_t cap = std::max<size_t>(dev_bytes * 2, 1 << 14);
    HIP_TRY(hipHostMalloc(&s.h, cap, hipHostMallocDefault));
    HIP_TRY(hipMalloc(&s.d, cap));
    s.cap = cap;
  }
  s.lay.d_cuts = static_cast<CutDesc*>(s.d);
  // a few workgroups per CU take the items round-robin; the grid is the smallest one that gives every workgroup the same number of items
  static const int64_t mb_slots = exp_env("HIPFEAT_MB_SLOTS") ? std::max(1, atoi(exp_env("HIPFEAT_MB_SLOTS"))) : 1792;
  const int64_t items = fill_items + s.res_blocks, per_wg = std::max<int64_t>(1, (items + mb_slots - 1) / mb_slots);
  const unsigned grid = (unsigned)std::max<int64_t>(1, (items + per_wg - 1) / per_wg);
  MbInlineArgs args;  // (header + 3.3 KB; only the used part of the blob is written)
  MbHeader& h = args.h;
  h.arena = d_arena;
  h.out = d_out;
  h.cuts_dst = static_cast<CutDesc*>(s.d);
  h.tables = nullptr;
  for (int k = 0; k < kMbKinds; ++k) h.kt[k] = bank->kt[k];
  h.num_cuts = (int32_t)batch;
  h.num_res = (int32_t)s.res.size();
  h.fill_items = (int32_t)fill_items;
  h.res_items = (int32_t)s.res_blocks;
  h.table_bytes = (int32_t)bytes;
  h.copy_descs = inl ? 1 : 0;
  h.feature_dim = F;
  h.pad_value = pad_value;
  h.feature_blocks = (int32_t)s.lay.total_blocks;
  h.quads_per_wg = s.lay.flat ? (int32_t)(s.lay.fpb / 4) : 0;
  s.lay.uniform_bpc = -1;  // (the map is always there, uniform or not)
  auto fill_blob = [&](unsigned char* dst) {
    if (res_bytes) std::memcpy(dst, s.res.data(), res_bytes);
    std::memcpy(dst + res_bytes, s.descs.data(), cut_bytes);
    std::memcpy(dst + res_bytes + cut_bytes, s.fill_first.data(), fill_bytes);
    std::memcpy(dst + res_bytes + cut_bytes + fill_bytes, s.rows.data(), row_bytes);
  };
  hipError_t e1 = hipSuccess;
  auto fill_map = [&](unsigned char* dst) {
    int32_t* m = reinterpret_cast<int32_t*>(dst);
    if (s.lay.flat) {  // (laid out by frame quads: the map build_descs computed)
      std::memcpy(m, s.lay.block_cut.data(), s.lay.block_cut.size() * sizeof(int32_t));
      return;
    }
    for (int64_t b = 0; b < batch; ++b) {
      const int64_t b0 = s.descs[(size_t)b].first_block, b1 = b + 1 < batch ? s.descs[(size_t)b + 1].first_block : s.lay.total_blocks;
      std::fill(m + b0, m + b1, (int32_t)b);
    }
  };
  if (dbg_skip == 2) {  // (experiments: no prep launch; the descriptor table of the feature launch still has to get there)
    std::memcpy(s.h, s.descs.data(), cut_bytes);
    fill_map(static_cast<unsigned char*>(s.h) + cut_bytes);
    HIP_TRY(hipMemcpyAsync(s.d, s.h, cut_bytes + map_bytes, hipMemcpyHostToDevice, st));
  } else if (inl) {
    fill_blob(args.blob);
    hipLaunchKernelGGL(minibatch_prep_inline_kernel, dim3(grid), dim3(256), 0, st, args);
    e1 = hipGetLastError();
  } else {
    unsigned char* hb = static_cast<unsigned char*>(s.h);
    std::memcpy(hb, s.descs.data(), cut_bytes);
    fill_map(hb + cut_bytes);
    fill_blob(hb + cut_bytes + map_bytes);
    HIP_TRY(hipMemcpyAsync(s.d, s.h, dev_bytes, hipMemcpyHostToDevice, st));
    h.tables = static_cast<const unsigned char*>(s.d) + cut_bytes + map_bytes;
    hipLaunchKernelGGL(minibatch_prep_kernel, dim3(grid), dim3(256), bytes <= (size_t)kMbLdsTableBytes ? bytes : 0, st, h);
    e1 = hipGetLastError();
  }
  hipfeat_status rc = HIPFEAT_OK;
  if (e1 != hipSuccess) rc = fail(HIPFEAT_ERR_HIP, "mini-batch prep launch failed: %s", hipGetErrorName(e1));
  if (rc == HIPFEAT_OK && s.lay.total_blocks > 0 && dbg_skip != 1) rc = launch(plan, &s.lay, d_arena, d_out, st);
  hipError_t e2 = hipEventRecord(s.ev, st);
  s.busy = (e2 == hipSuccess);
  return rc;
}

// ---- bulk save path: per-batch host work of the offline driver (host_bulk.hpp) -------------------------------------------------
struct hipfeat_archive {
  hipfeat::ArchiveFiles files;
  std::mutex mu;
};

extern "C" HIPFEAT_API hipfeat_status hipfeat_archive_open(const char* const* h_paths, int32_t num_files, int32_t append, hipfeat_archive** out) {
  if (!out) return fail(HIPFEAT_ERR_INVALID, "archive pointer is NULL");
  *out = nullptr;
  if (!h_paths || num_files < 1 || num_files > 64) return fail(HIPFEAT_ERR_INVALID, "an archive has 1 ... 64 files");
  hipfeat_archive* a = new (std::nothrow) hipfeat_archive();
  if (!a) return fail(HIPFEAT_ERR_INVALID, "out of host memory");
  for (int32_t k = 0; k < num_files; ++k) {
    const int fd = h_paths[k] ? ::open(h_paths[k], O_CREAT | O_WRONLY | O_CLOEXEC | (append ? 0 : O_TRUNC), 0644) : -1;
    const off_t end = fd >= 0 ? ::lseek(fd, 0, SEEK_END) : (off_t)-1;
    if (fd < 0 || end < 0) {
      const int e = errno;
      if (fd >= 0) ::close(fd);
      for (int f : a->files.fds) ::close(f);
      delete a;
      return fail(HIPFEAT_ERR_INVALID, "cannot open archive file %d (%s): %s", k, h_paths[k] ? h_paths[k] : "NULL", strerror(e));
    }
    a->files.fds.push_back(fd);
    a->files.size.push_back((int64_t)end);
  }
  if (num_files > 1) a->files.pool = new hipfeat::WorkPool(num_files - 1, "hipfeat-stripe");
  *out = a;
  return HIPFEAT_OK;
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_archive_append(hipfeat_archive* a, const void* h_matrix, int64_t batch, const int64_t* h_num_frames,
                                                             int32_t cols, int32_t bytes_per_value, int32_t* h_file, int64_t* h_byte_offset) {
  if (!a || !h_num_frames || (batch > 0 && !h_matrix) || batch < 0) return fail(HIPFEAT_ERR_INVALID, "NULL argument");
  if (cols <= 0 || (bytes_per_value != 2 && bytes_per_value != 4)) return fail(HIPFEAT_ERR_INVALID, "rows are %d values of %d bytes: need cols > 0 and 2- or 4-byte values", cols, bytes_per_value);
  for (int64_t i = 0; i < batch; ++i)
    if (h_num_frames[i] < 0) return fail(HIPFEAT_ERR_INVALID, "cut %lld: %lld frames", (long long)i, (long long)h_num_frames[i]);
  std::lock_guard<std::mutex> lk(a->mu);
  int errfile = 0;
  const int e = a->files.append(static_cast<const char*>(h_matrix), batch, h_num_frames, (int64_t)cols * bytes_per_value, bytes_per_value == 2, h_file,
                                h_byte_offset, &errfile);
  if (e == hipfeat::kErrNotFinite16)
    return fail(HIPFEAT_ERR_INVALID, "the batch holds values that are not finite in binary16 (|x| > 65504, inf or nan): nothing was written");
  if (e) return fail(HIPFEAT_ERR_INVALID, "write to archive file %d failed: %s", errfile, strerror(e));
  return HIPFEAT_OK;
}

extern "C" HIPFEAT_API int64_t hipfeat_archive_size(const hipfeat_archive* a, int32_t file) {
  return (a && file >= 0 && file < (int32_t)a->files.size.size()) ? a->files.size[(size_t)file] : -1;
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_archive_close(hipfeat_archive* a) {
  if (!a) return HIPFEAT_OK;
  int bad = 0;
  for (int fd : a->files.fds)
    if (::close(fd) != 0) bad = errno;
  delete a->files.pool;
  delete a;
  return bad ? fail(HIPFEAT_ERR_INVALID, "closing an archive file failed: %s", strerror(bad)) : HIPFEAT_OK;
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_manifest_lines(const char* h_heads, const int64_t* h_head_offsets, const char* h_tails,
                                                             const int64_t* h_tail_offsets, int64_t batch, const int64_t* h_num_frames,
                                                             const int64_t* h_expected_frames, const char* h_mids, const int64_t* h_mid_offsets,
                                                             int32_t num_files, const int32_t* h_file, const int64_t* h_byte_offset, int32_t cols,
                                                             int32_t bytes_per_value, char* h_out, int64_t out_capacity, int64_t* h_out_bytes) {
  if (!h_heads || !h_head_offsets || !h_tails || !h_tail_offsets || !h_num_frames || !h_mids || !h_mid_offsets || !h_byte_offset || !h_out || !h_out_bytes)
    return fail(HIPFEAT_ERR_INVALID, "NULL argument");
  if (batch < 0 || num_files < 1 || cols <= 0 || (bytes_per_value != 2 && bytes_per_value != 4)) return fail(HIPFEAT_ERR_INVALID, "bad batch / file / row arguments");
  hipfeat::BulkError err;
  const int64_t n = hipfeat::manifest_lines(h_heads, h_head_offsets, h_tails, h_tail_offsets, batch, h_num_frames, h_expected_frames, h_mids, h_mid_offsets,
                                            num_files, h_file, h_byte_offset, cols, bytes_per_value, h_out, out_capacity, &err);
  if (n < 0) {
    *h_out_bytes = err.index < 0 ? err.a : 0;
    if (err.index >= 0 && err.what[0] == 'f' && err.what[1] == 'r')
      return fail(HIPFEAT_ERR_INVALID, "cut %lld of the batch: %lld frames extracted, its manifest states %lld (the frame-count contract of validate_features)",
                  (long long)err.index, (long long)err.a, (long long)err.b);
    return fail(HIPFEAT_ERR_INVALID, "%s (cut %lld: %lld vs %lld)", err.what, (long long)err.index, (long long)err.a, (long long)err.b);
  }
  *h_out_bytes = n;
  return HIPFEAT_OK;
}

#include "host_pipeline.hpp"

#ifdef HIPFEAT_PHASE_TIMERS
// experiment builds only: install the per-wave phase-clock buffer ([waves][8] uint64, device memory)
extern "C" HIPFEAT_API int hipfeat_debug_set_phase_buffer(unsigned long long* d_buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(hipfeat::g_phase_buf), &d_buf, sizeof(d_buf)) == hipSuccess ? 0 : 1;
}
#endif
