/*
 * hipfeat.h -- C ABI of libhipfeat.so: batched Kaldi-style audio feature extraction
 * (spectrogram / log-spectrogram / log-mel filterbank / MFCC) on AMD MI355X (gfx950).
 *
 * This is the drop-in boundary for ONE path of lhotse: the arithmetic behind
 *   FeatureExtractor.extract / extract_batch   (lhotse/features/base.py:75-82, :152-222)
 * as implemented by the torch layers
 *   Wav2Win -> Wav2FFT -> Wav2Spec / Wav2LogSpec / Wav2LogFilterBank / Wav2MFCC
 *                                              (lhotse/features/kaldi/layers.py:59-724)
 * and driven by _extract_batch                 (lhotse/features/kaldi/extractors.py:485-554).
 *
 * The reference has no native code and therefore no FFI for this path; its FFI
 * convention elsewhere is ctypes.CDLL + integer status codes
 * (lhotse/tools/libsox.py:74-117).  Every entry point below is plain C: opaque
 * handles, raw pointers, sizes; no torch / Python types.  It is loadable with
 * ctypes or cffi (see INTEGRATION.md for the binding a lhotse maintainer would add).
 *
 * Conventions
 *   - every function returns hipfeat_status (0 == OK) unless stated otherwise;
 *     hipfeat_last_error() returns a thread-local message for the last failure;
 *   - "d_" pointers are device (HBM) pointers on the plan's device, "h_" pointers are host;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is
 *     enqueued asynchronously on it, nothing synchronises the device;
 *   - no global state; safe to call with the Python GIL released; a plan may be used
 *     from several host threads as long as they use different streams.
 */
#ifndef HIPFEAT_H_
#define HIPFEAT_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIPFEAT_ABI_VERSION 5

#if defined(HIPFEAT_BUILD)
#define HIPFEAT_API __attribute__((visibility("default")))
#else
#define HIPFEAT_API
#endif

typedef enum hipfeat_status {
  HIPFEAT_OK = 0,
  HIPFEAT_ERR_INVALID = 1,     /* bad argument / inconsistent config                        */
  HIPFEAT_ERR_HIP = 2,         /* a HIP runtime call failed (message has the hipError name) */
  HIPFEAT_ERR_UNSUPPORTED = 3, /* valid in the reference but not implemented here           */
  HIPFEAT_ERR_TOO_SHORT = 4    /* waveform shorter than the reflect padding needs; the
                                  reference raises here too (layers.py:759-764)             */
} hipfeat_status;

/* Which layer of lhotse/features/kaldi/layers.py the plan reproduces. */
typedef enum hipfeat_kind {
  HIPFEAT_SPECTROGRAM = 0,     /* Wav2Spec            layers.py:336-402 */
  HIPFEAT_LOG_SPECTROGRAM = 1, /* Wav2LogSpec         layers.py:405-473 */
  HIPFEAT_FBANK = 2,           /* Wav2LogFilterBank   layers.py:476-578 */
  HIPFEAT_MFCC = 3,            /* Wav2MFCC            layers.py:581-724 */
  /* "next" row: Whisper log-mel, log_mel_spectrogram lhotse/features/whisper_fbank.py:17-85.
   * Implies: centred frames (frame t covers samples [t*shift - N/2, t*shift + N/2), torch.stft(center=True)),
   * "reflect" edges without repeating the edge sample, fft_length == frame_length (any size, direct DFT),
   * S / shift computed frames, log10(max(mel, mel_floor)) clamped to (per-cut max - 8), then (x + 4) / 4,
   * and (S + shift/2) / shift output rows, the extra one (if any) all zeros.  window / mel as for HIPFEAT_FBANK.
   * MEMORY-MODEL NOTE (gfx950 only, by construction of this library): the per-cut clamp is finished by whichever workgroup of the
   * launch completes the cut last; the hand-off between workgroups uses agent-scope write-through (sc1) stores + vmcnt(0) in front
   * of a relaxed agent-scope counter increment, and sc1 loads behind it -- the ordering the gfx950 memory pipeline gives, NOT a
   * release / acquire pair of the HIP memory model (that pair costs an L2 write-back + invalidate per workgroup: 1.9 -> 12.7 ms per
   * 4000 cuts, measured in round 3).  The library is built for --offload-arch=gfx950 alone and refuses other devices at plan creation,
   * so no caller can meet this path on hardware where the assumption does not hold; a launch that faults leaves the per-cut counters
   * of its layout armed -- destroy the layout (or plan) after a HIPFEAT_ERR_HIP from a Whisper extraction instead of re-using it. */
  HIPFEAT_WHISPER = 4,
  /* librosa-style log-mel (lhotse/features/librosa_fbank.py:66-137): the arithmetic of HIPFEAT_FBANK (set use_fft_mag = 1
   * for librosa's |X|) on centred frames with "reflect" edges (librosa.stft center=True, pad_mode="reflect"), log10 instead of
   * ln, (S + shift/2) / shift rows.  window = periodic hann of fft_length samples, mel = slaney filters (caller's values). */
  HIPFEAT_LIBROSA_FBANK = 5
} hipfeat_kind;

/*
 * Scalar options.  Field meaning == the constructor arguments of the layers above
 * (layers.py:80-94, :247-261, :494-514, :599-621), already converted to samples.
 * The float32 constant arrays (window, mel matrix, DCT, lifter) are passed separately
 * to hipfeat_plan_create so that the caller -- who computes them with the reference's
 * own formulae -- owns their exact values.
 */
typedef struct hipfeat_config {
  int32_t struct_size;      /* = sizeof(hipfeat_config); ABI guard                              */
  int32_t kind;             /* hipfeat_kind                                                     */
  int32_t frame_length;     /* N  = floor(frame_length_s * sampling_rate)    layers.py:114     */
  int32_t frame_shift;      /* floor(frame_shift_s * sampling_rate)          layers.py:116     */
  int32_t fft_length;       /* next_power_of_2(N) or N                       layers.py:265     */
  int32_t num_filters;      /* M (mel bins);   0 for (log-)spectrogram                          */
  int32_t num_ceps;         /* C (MFCC only);  0 otherwise                                      */
  int32_t snip_edges;       /* layers.py:747-753                                                */
  int32_t remove_dc_offset; /* layers.py:155-157                                                */
  int32_t use_energy;       /* layers.py:575-576 (fbank: prepended column), :399-400, :470-471;
                               MFCC: log-energy replaces C0 (Kaldi; the intent of :721-722)     */
  int32_t raw_energy;       /* layers.py:161 vs :183                                            */
  int32_t use_fft_mag;      /* |X| instead of |X|^2                          layers.py:387-390 */
  int32_t apply_lifter;     /* cepstral_lifter > 0                           layers.py:717     */
  float preemph_coeff;      /* 0 disables                                    layers.py:165     */
  float energy_floor;       /* log-energy floored at log(energy_floor) if >0 layers.py:864     */
  float mel_floor;          /* eps of max(mel, eps).log() = 1.1920929e-07    layers.py:536,572 */
  float log_offset;         /* 1e-15 added before log in Wav2LogSpec         layers.py:467     */
  float dither;             /* must be 0 (layers.py:191-193 draws torch.randn: the host adds it)  */
  int32_t batch_hop;        /* round(frame_shift_s * sampling_rate): the hop with which
                               compute_num_frames_from_samples (lhotse/utils.py:424-434) counts the rows
                               item b keeps of a zero-padded batch row (padded_len given); differs from
                               frame_shift for fractional hops (12.5 ms @ 22.05 kHz: 276 vs 275).
                               0 = frame_shift                                                   */
} hipfeat_config;

typedef struct hipfeat_plan hipfeat_plan;     /* constants + kernel selection, per (device, config) */
typedef struct hipfeat_layout hipfeat_layout; /* device-resident description of one batch shape     */

/* ---- library ------------------------------------------------------------------- */
HIPFEAT_API int32_t hipfeat_abi_version(void);
HIPFEAT_API const char* hipfeat_last_error(void);
HIPFEAT_API hipfeat_status hipfeat_device_count(int32_t* count);

/* ---- pure host helpers (no GPU needed) ------------------------------------------ */
/* Frame count of one waveform: layers.py:747-753; for snip_edges == 0 it equals
 * compute_num_frames_from_samples (lhotse/utils.py:424-434), the contract that
 * validate_features asserts (lhotse/qa.py:286-311). */
HIPFEAT_API int64_t hipfeat_num_frames(int64_t num_samples, int32_t frame_length, int32_t frame_shift, int32_t snip_edges);
/* HIPFEAT_OK, or HIPFEAT_ERR_TOO_SHORT when reflect padding is impossible:
 * the reflection is taken on a row of `padded_len` samples (== num_samples for a
 * single waveform). */
HIPFEAT_API hipfeat_status hipfeat_check_length(int64_t padded_len, int32_t frame_length, int32_t frame_shift, int32_t snip_edges);

/* ---- plan ------------------------------------------------------------------------ */
/* window[frame_length]; mel[(fft_length/2+1) * num_filters] row-major (bin, filter) as in
 * Wav2LogFilterBank._fb (layers.py:553,563); dct[num_filters * num_ceps] row-major as in
 * Wav2MFCC._dct (layers.py:697-706); lifter[num_ceps] (layers.py:681-695).  Unused arrays
 * may be NULL.  All are HOST float32 arrays, copied to the device. */
HIPFEAT_API hipfeat_status hipfeat_plan_create(const hipfeat_config* cfg, const float* h_window, const float* h_mel,
                                   const float* h_dct, const float* h_lifter, int32_t device,
                                   hipfeat_plan** plan);
HIPFEAT_API hipfeat_status hipfeat_plan_destroy(hipfeat_plan* plan);
/* Columns written per frame: M (+1 with use_energy) | C | fft/2+1. */
HIPFEAT_API int32_t hipfeat_plan_feature_dim(const hipfeat_plan* plan);
/* Human-readable name of the kernel variant the plan dispatches to (e.g. "generic",
 * "fft512"); tests use it to assert that the specialised path is the one exercised. */
HIPFEAT_API const char* hipfeat_plan_kernel_name(const hipfeat_plan* plan);

/* ---- layout: the shape of a batch ------------------------------------------------ */
/*
 * A batch is `batch` cuts.  Cut b occupies h_wave_offsets[b] .. + h_num_samples[b] (in
 * float32 elements) of the waveform buffer -- this covers both the packed form
 * (offsets = prefix sums) and the padded (B, Smax) form (offsets = b * Smax).
 *
 * h_padded_len (nullable): the row length the edge reflection is taken on.
 *   NULL / == num_samples : every cut is reflected on its own, as Fbank.extract
 *                           (extractors.py:92-115) and kaldifeat do;
 *   == max(num_samples)   : reproduces _extract_batch (extractors.py:531-537): shorter
 *                           items read zeros past their end, and only
 *                           compute_num_frames_from_samples(num_samples) rows are produced.
 *
 * Output: cut b's features are written to rows h_out_rows[b] .. + num_frames(b) of a
 * row-major matrix with `out_row_stride` floats per row (>= feature_dim).  h_out_rows
 * NULL = packed back to back.  This covers the packed (sum T_b, F) form and the padded
 * (B, Tmax, F) form (out_rows = b * Tmax).
 */
HIPFEAT_API hipfeat_status hipfeat_layout_create(const hipfeat_plan* plan, int64_t batch, const int64_t* h_wave_offsets,
                                     const int64_t* h_num_samples, const int64_t* h_padded_len,
                                     const int64_t* h_out_rows, int64_t out_row_stride, void* stream,
                                     hipfeat_layout** layout);
HIPFEAT_API hipfeat_status hipfeat_layout_destroy(hipfeat_layout* layout);
HIPFEAT_API int64_t hipfeat_layout_total_frames(const hipfeat_layout* layout);
/* Writes the per-cut frame counts into h_num_frames[batch]. */
HIPFEAT_API hipfeat_status hipfeat_layout_num_frames(const hipfeat_layout* layout, int64_t* h_num_frames);

/* ---- the hot path ---------------------------------------------------------------- */
/* One asynchronous pass over the batch described by `layout`: d_wave -> d_out. */
HIPFEAT_API hipfeat_status hipfeat_extract_layout(const hipfeat_plan* plan, const hipfeat_layout* layout,
                                      const float* d_wave, float* d_out, void* stream);

/* Convenience: build a transient layout from host arrays and run it (same semantics). */
HIPFEAT_API hipfeat_status hipfeat_extract(const hipfeat_plan* plan, const float* d_wave, const int64_t* h_wave_offsets,
                               const int64_t* h_num_samples, const int64_t* h_padded_len, int64_t batch,
                               float* d_out, const int64_t* h_out_rows, int64_t out_row_stride, void* stream);

/* Host-memory form: h_wave / h_out are HOST buffers (pinned or pageable); the library
 * stages them through the device (H2D, kernel, D2H) on `stream` and waits for completion.
 * `wave_elems` / `out_elems` are the total buffer sizes in float32 elements. */
HIPFEAT_API hipfeat_status hipfeat_extract_host(const hipfeat_plan* plan, const float* h_wave, int64_t wave_elems,
                                    const int64_t* h_wave_offsets, const int64_t* h_num_samples,
                                    const int64_t* h_padded_len, int64_t batch, float* h_out,
                                    int64_t out_elems, const int64_t* h_out_rows, int64_t out_row_stride,
                                    void* stream);


/* ---- "next" row: fused collation of the output, int16 PCM input ------------------- */
/*
 * hipfeat_extract + collate_matrices(features, padding_value) in one call
 * (lhotse/dataset/input_strategies.py:458-462, lhotse/dataset/collation.py:506-535): d_out is a dense
 * (batch, rows_per_cut, feature_dim) float32 tensor; cut b's frames go to d_out[b, :T_b] and the rows
 * [T_b, rows_per_cut) are filled with pad_value (LOG_EPSILON in lhotse).  T_b is returned in h_num_frames
 * (may be NULL).  Fails if some T_b > rows_per_cut.
 */
HIPFEAT_API hipfeat_status hipfeat_extract_collated(const hipfeat_plan* plan, const float* d_wave, const int64_t* h_wave_offsets,
                                        const int64_t* h_num_samples, const int64_t* h_padded_len, int64_t batch,
                                        float* d_out, int64_t rows_per_cut, float pad_value, int64_t* h_num_frames,
                                        void* stream);
/* int16 PCM -> float32 in [-1, 1): x / 32768, exactly what the audio backends hand to the reference
 * (so features are bit-identical to the float32 path); lets the host send half the bytes over PCIe. */
HIPFEAT_API hipfeat_status hipfeat_pcm16_to_float(const int16_t* d_pcm, float* d_wave, int64_t num_samples, void* stream);
/* float32 -> IEEE binary16 (round to nearest even) on the device, in front of the device -> host copy of a feature batch that is going to
 * be STORED in half precision: the reference's default feature storage is lossy as well (lilcom, lhotse/features/io.py:981-1061,
 * features/compression.py:18-37: the fixture it ships is exact to 2^-6); binary16 keeps log-domain features (|x| < 32) to 2^-6 ... 2^-7
 * absolute and halves both the PCIe traffic and the file.  lilcom's own bit stream is third-party and not reproduced.  No plan: launches
 * on the calling thread's current device. */
HIPFEAT_API hipfeat_status hipfeat_float_to_half(const float* d_in, uint16_t* d_out, int64_t n, void* stream);

/* ---- "next" row: post-feature transforms on the collated (B, T, F) batch ------- */
/* GlobalMVN.forward: (x - means) / stds, and .inverse: x * stds + means, over `rows` rows of `feature_dim` floats
 * (lhotse/dataset/signal_transforms.py:50-60).  IEEE single operations in the reference's order: bit-identical. */
HIPFEAT_API hipfeat_status hipfeat_global_mvn(const float* d_in, float* d_out, const float* d_means, const float* d_stds, int64_t rows,
                                              int64_t feature_dim, int inverse, void* stream);
/*
 * SpecAugment._forward_single for every sequence of a batch in two launches
 * (lhotse/dataset/signal_transforms.py:173-266): d_out = clone of d_in with
 *   - every warp segment time-warped: rows [0, center) of the segment are resampled to [0, warped) and rows
 *     [center, num_frames) to [warped, num_frames) with torch's bicubic interpolation (time_warp, :338-371;
 *     F.interpolate(mode="bicubic", align_corners=False)).  `center` and `warped` are the two np.random.randint draws;
 *     segments of one sequence must not overlap (the reference applies them one after another);
 *   - every mask region set to the mean of ITS sequence (all num_frames * feature_dim values, after warping):
 *     axis 1 = frames [begin, end), axis 2 = feature bins [begin, end) (mask_along_axis_optimized, :297-335).
 * The random draws stay with the caller (the host mirror makes them with the reference's RNG calls in the reference's order).
 * Descriptors are host arrays; the call is asynchronous on `stream`.  d_in and d_out must not alias.
 * Like hipfeat_pcm16_to_float and hipfeat_global_mvn it has no plan: it launches on the calling thread's current device, which must
 * be the one that owns the buffers and the stream.
 */
typedef struct hipfeat_warp_segment {
  int32_t sequence, start, num_frames, center, warped;
} hipfeat_warp_segment;
typedef struct hipfeat_mask {
  int32_t sequence, axis, begin, end;
} hipfeat_mask;
HIPFEAT_API hipfeat_status hipfeat_specaug(const float* d_in, float* d_out, int64_t batch, int64_t num_frames, int64_t feature_dim,
                                           const hipfeat_warp_segment* h_segments, int64_t num_segments, const hipfeat_mask* h_masks,
                                           int64_t num_masks, void* stream);

/* ---- "next" row: speed perturbation = polyphase sinc resampling --------------- */
/*
 * Replaces ResampleTensor (lhotse/augmentation/resample.py:42-142, :284-315) as used by
 * Speed.__call__ (lhotse/augmentation/torchaudio.py:37-42).  `orig_freq` / `new_freq` are the
 * gcd-reduced rates; h_kernel is the float32 filter bank [new_freq][2*width + orig_freq] computed by
 * the caller with the reference's formula (resample.py:184-281), so its values are the caller's.
 */
typedef struct hipfeat_resampler hipfeat_resampler;
HIPFEAT_API hipfeat_status hipfeat_resampler_create(int32_t orig_freq, int32_t new_freq, int32_t width, const float* h_kernel,
                                        int32_t device, hipfeat_resampler** resampler);
HIPFEAT_API hipfeat_status hipfeat_resampler_destroy(hipfeat_resampler* resampler);
/* ceil(new * num_samples / orig), evaluated like the reference (in float32, resample.py:309). */
HIPFEAT_API int64_t hipfeat_resampled_length(int64_t num_samples, int32_t orig_freq, int32_t new_freq);
/* Batch of cuts: cut b = d_in[h_in_offsets[b] .. + h_num_samples[b]) -> d_out[h_out_offsets[b] .. + resampled_length). */
HIPFEAT_API hipfeat_status hipfeat_resample(const hipfeat_resampler* resampler, const float* d_in, const int64_t* h_in_offsets,
                                const int64_t* h_num_samples, int64_t batch, float* d_out, const int64_t* h_out_offsets,
                                void* stream);

/* ---- on-the-fly mini-batch: mixed-factor speed perturbation + collated extraction in TWO launches ------------- */
/*
 * BASELINE configs[4].  What OnTheFlyFeatures does per mini-batch on the CPU -- Speed per cut inside Recording.load_audio
 * (lhotse/dataset/cut_transforms/perturb_speed.py:8-47, lhotse/augmentation/torchaudio.py:37-42, augmentation/resample.py:284-315),
 * extract_batch, collate_matrices with LOG_EPSILON (lhotse/dataset/input_strategies.py:410-462, dataset/collation.py:506-535) --
 * on a packed mini-batch that is resident in ONE device buffer (the "arena": the cuts at h_offsets / h_num_samples in its front
 * part, free space from tail_start on), as a pair of launches: (1) every perturbed cut, whatever its factor, is resampled into the
 * arena's tail (one workgroup = 256 hops of one cut, the polyphase bank selected per cut), the padding rows of the collated tensor are
 * filled and the descriptor table of launch (2) is put in place -- for mini-batches of up to ~100 cuts the tables travel in the kernel
 * arguments, i.e. there is no host -> device copy in front of the launches at all; (2) the feature kernel of `plan` reads every cut
 * where it now lies (unperturbed cuts are never copied) and writes it into its slot of the dense (batch, rows_per_cut, feature_dim)
 * tensor.  Results are bit-identical to hipfeat_resample per factor + hipfeat_extract_collated.
 *
 * A bank = the resamplers a mini-batch may refer to (h_bank_index[b] = index into the bank, -1 = unperturbed); they must be of the
 * compile-time ratios 9:10 and 11:10 (speed 0.9 / 1.1, width 7: the factors of Kaldi-style three-way speed perturbation) -- anything
 * else: HIPFEAT_ERR_UNSUPPORTED, use hipfeat_resample per factor -- and must outlive the bank.
 *
 * hipfeat_minibatch_plan is host arithmetic only: where each perturbed cut goes (h_out_offsets), how long it is (h_out_num_samples;
 * ceil(new * n / orig) as hipfeat_resampled_length, capped by h_max_samples[b] >= 0 when given: lhotse truncates a perturbed cut to
 * the sample count its manifest states, lhotse/audio/recording.py:1058-1060), its frame count (h_num_frames), and in h_info[4] =
 * {ticket, floats the arena must hold, largest frame count, rows of the output} -- what the caller needs to allocate the arena and the
 * output.  zero_pad_batch != 0 frames every cut on a row of the longest cut's length (edge_rule "batch_zero_pad", _extract_batch's rule).
 * SEVERAL mini-batches per launch pair (a prefetching loader: fewer, larger launches fill 256 CUs better than one 600 s mini-batch can):
 * num_groups > 1 and h_group_sizes[k] = cuts of mini-batch k (consecutive cuts; the sizes add up to `batch`).  Every mini-batch then is
 * its own dense (B_k, T_k, feature_dim) tensor, T_k = its longest cut, the tensors back to back in d_out: h_group_rows[2k] = first row,
 * h_group_rows[2k + 1] = T_k; h_info[3] = rows in total.  num_groups <= 1 (h_group_sizes may be NULL): one mini-batch.
 * hipfeat_minibatch_run enqueues the two launches of a planned mini-batch on `stream`; one mini-batch: rows_per_cut >= h_info[2] (the
 * caller may want more rows than the longest cut has); grouped plans: rows_per_cut = -1, d_out holds h_info[3] rows.  Up to 16 plans may
 * be outstanding per bank; a bank may be shared by threads (calls are serialised inside).
 */
typedef struct hipfeat_speed_bank hipfeat_speed_bank;
HIPFEAT_API hipfeat_status hipfeat_speed_bank_create(const hipfeat_resampler* const* resamplers, int32_t num_resamplers,
                                                     hipfeat_speed_bank** bank);
HIPFEAT_API hipfeat_status hipfeat_speed_bank_destroy(hipfeat_speed_bank* bank);
HIPFEAT_API hipfeat_status hipfeat_minibatch_plan(hipfeat_speed_bank* bank, const hipfeat_plan* plan, int64_t batch, const int64_t* h_offsets,
                                                  const int64_t* h_num_samples, const int32_t* h_bank_index, const int64_t* h_max_samples,
                                                  int64_t tail_start, int32_t zero_pad_batch, int64_t num_groups, const int64_t* h_group_sizes,
                                                  int64_t* h_out_offsets, int64_t* h_out_num_samples, int64_t* h_num_frames,
                                                  int64_t* h_group_rows, int64_t* h_info);
HIPFEAT_API hipfeat_status hipfeat_minibatch_run(hipfeat_speed_bank* bank, int64_t ticket, float* d_arena, int64_t arena_floats, float* d_out,
                                                 int64_t rows_per_cut, float pad_value, void* stream);

/* ---- bulk save path: the per-batch host work of the offline driver ------------------------------- */
/*
 * What lhotse's _save_worker does per CUT in the interpreter behind compute_and_store_features_batch (lhotse/cut/set.py:2307-2363:
 * FeaturesWriter.write of one matrix -- lhotse/features/io.py:499-525 --, a Features object, validate_features, fastcopy(cut),
 * to_dict, json) is done per BATCH here, in plain host code that runs with the GIL released; no device is touched.
 *
 * hipfeat_archive_*: an append-only archive of feature rows striped over num_files flat files (1 = the single-file "hip_archive" of
 * lhotse_amd/storage.py).  hipfeat_archive_append writes the packed (sum h_num_frames, cols) matrix of one batch exactly as it left the
 * device (float32: bytes_per_value 4; binary16: 2): the batch is cut into num_files consecutive runs of whole cuts of about equal
 * bytes, run k is appended to file k by its own thread (writers to different files do not share an inode's page-cache locks);
 * h_file[b] / h_byte_offset[b] say where the rows of cut b begin.  The storage key of a cut is "<byte offset>:<rows>:<cols>[:f16]".
 *
 * hipfeat_manifest_lines: the JSONL lines of one batch.  Line b = head_b + mid[h_file[b]] + key_b + tail_b + "\n", where head_b / tail_b
 * are the two halves of the cut's own serialisation up to / from the storage fields (made by the caller where the cut was loaded --
 * lhotse's loader workers -- because they do not depend on the extraction), mid[k] = `<JSON-escaped path of file k>", "storage_key": "`
 * and key_b the key above.  heads / tails / mids are concatenated byte strings with (n + 1) int64 offsets each.  h_expected_frames
 * (optional; entries < 0 are skipped) is the frame count each manifest half states: a mismatch with h_num_frames is the frame-count
 * contract validate_features asserts (lhotse/qa.py:286-301) and fails the call before anything is written.  h_out needs
 * sum(len(head_b) + len(tail_b) + longest mid + 65) bytes; on "too small" *h_out_bytes holds that number.
 */
typedef struct hipfeat_archive hipfeat_archive;
HIPFEAT_API hipfeat_status hipfeat_archive_open(const char* const* h_paths, int32_t num_files, int32_t append, hipfeat_archive** archive);
HIPFEAT_API hipfeat_status hipfeat_archive_append(hipfeat_archive* archive, const void* h_matrix, int64_t batch, const int64_t* h_num_frames,
                                                  int32_t cols, int32_t bytes_per_value, int32_t* h_file, int64_t* h_byte_offset);
HIPFEAT_API int64_t hipfeat_archive_size(const hipfeat_archive* archive, int32_t file); /* bytes in file `file`, -1 if out of range */
HIPFEAT_API hipfeat_status hipfeat_archive_close(hipfeat_archive* archive);
HIPFEAT_API hipfeat_status hipfeat_manifest_lines(const char* h_heads, const int64_t* h_head_offsets, const char* h_tails,
                                                  const int64_t* h_tail_offsets, int64_t batch, const int64_t* h_num_frames,
                                                  const int64_t* h_expected_frames, const char* h_mids, const int64_t* h_mid_offsets,
                                                  int32_t num_files, const int32_t* h_file, const int64_t* h_byte_offset, int32_t cols,
                                                  int32_t bytes_per_value, char* h_out, int64_t out_capacity, int64_t* h_out_bytes);

/* ---- the offline driver's extraction step, one asynchronous call per batch of HOST waveforms -------------------- */
/*
 * Behind compute_and_store_features_batch the extractor is handed a list of host tensors per batch and its result goes to a save
 * thread (lhotse/cut/set.py:2365-2404).  A host pipeline does that step inside the library: hipfeat_host_pipeline_submit packs the
 * cuts (h_items[b] = pointer to h_num_samples[b] float32 samples, or int16 PCM when pcm16 != 0; pageable or page-locked) into
 * page-locked staging with `copy_threads` persistent host threads, and ENQUEUES, chunk by chunk on two private streams, the upload,
 * [hipfeat_pcm16_to_float,] the plan's feature launch (per-item reflect edges; zero_pad_batch != 0: edge_rule "batch_zero_pad"),
 * [hipfeat_float_to_half when half_out != 0] and the download into a page-locked result buffer owned by the pipeline.  It returns
 * without waiting for the device: *h_out is where the packed (*h_out_rows, feature_dim) matrix WILL be (float32, or binary16 with
 * half_out), h_num_frames[b] the frame counts (known at once), *ticket the handle.  hipfeat_host_pipeline_wait blocks until that
 * batch's features are in *h_out; hipfeat_host_pipeline_release gives the buffer back (after which *h_out may be overwritten by a later
 * batch).  Up to 64 results may be outstanding; submitting batch n + 1 before waiting for batch n is the point: its packing and
 * upload overlap batch n's download.  One pipeline per plan and calling thread; wait / release may come from another thread.
 */
typedef struct hipfeat_host_pipeline hipfeat_host_pipeline;
HIPFEAT_API hipfeat_status hipfeat_host_pipeline_create(const hipfeat_plan* plan, int32_t copy_threads, hipfeat_host_pipeline** pipeline);
HIPFEAT_API hipfeat_status hipfeat_host_pipeline_destroy(hipfeat_host_pipeline* pipeline);
HIPFEAT_API hipfeat_status hipfeat_host_pipeline_submit(hipfeat_host_pipeline* pipeline, const void* const* h_items, const int64_t* h_num_samples,
                                                        int64_t batch, int32_t pcm16, int32_t zero_pad_batch, int32_t half_out,
                                                        int64_t* h_num_frames, void** h_out, int64_t* h_out_rows, int64_t* ticket);
HIPFEAT_API hipfeat_status hipfeat_host_pipeline_wait(hipfeat_host_pipeline* pipeline, int64_t ticket);
HIPFEAT_API hipfeat_status hipfeat_host_pipeline_release(hipfeat_host_pipeline* pipeline, int64_t ticket);
/* The pipeline thread's own clock since creation, h_stats[4] = {nanoseconds busy with batches, of those: packing into page-locked
 * staging, of those: waiting for a staging set's previous uploads / downloads (back-pressure from PCIe and the device), batches}:
 * lets a driver say which stage binds its run. */
HIPFEAT_API hipfeat_status hipfeat_host_pipeline_stats(const hipfeat_host_pipeline* pipeline, int64_t* h_stats);
/*
 * ABI v5: upload straight out of the loader's memory.  What lhotse's driver receives per batch is whatever its DataLoader delivered
 * (lhotse/cut/set.py:2374-2398): pageable arrays, which the pipeline above first copies into page-locked staging -- measured as the
 * largest single CPU consumer of the offline path once the loader keeps up (4-5 of 14 busy CPUs, profiles/r06_ring_loader_ab.txt).
 * hipfeat_host_register page-locks [ptr, ptr + bytes) of the CALLER's memory for DMA by `device` (any mapping: the slots of a
 * shared-memory ring that loader processes fill); hipfeat_host_unregister(ptr) undoes it (before the memory is unmapped; never while a
 * batch that reads it is outstanding).  hipfeat_host_pipeline_submit recognises a batch whose cuts all lie in ONE registered range, in
 * ascending order, each on a 16-byte boundary and (nearly) back to back, and uploads that span with one copy: no staging, no packing
 * threads.  Everything else about the call is unchanged (the cuts' memory must stay untouched until _wait returns, as before).
 * hipfeat_host_pipeline_direct_batches = how many batches of the pipeline went that way so far (-1: NULL pipeline).
 */
HIPFEAT_API hipfeat_status hipfeat_host_register(int32_t device, void* ptr, int64_t bytes);
HIPFEAT_API hipfeat_status hipfeat_host_unregister(void* ptr);
HIPFEAT_API int64_t hipfeat_host_pipeline_direct_batches(const hipfeat_host_pipeline* pipeline);

#ifdef __cplusplus
}
#endif
#endif /* HIPFEAT_H_ */
