/* libmfcc_hip -- C ABI of the MI355X (gfx950) MFCC engine.
 *
 * The reference (jvargeaux/speech_to_text) has no FFI for this path: its boundary is the Python
 * keyword call
 *     librosa.feature.mfcc(y=samples.numpy(), n_fft=2048, hop_length=512, n_mfcc=13)
 * at /root/reference/src/preprocess.py:134 (also :123), driven once per utterance by
 * SpeechToTextTrainer.process_audio (/root/reference/src/preprocess.py:131-144) from the loop in
 * SpeechToTextTrainer.preprocess (:146-167).  This header is the C-level equivalent of that
 * call for a ragged BATCH of utterances; the Python host layer (speech_to_text_amd/) binds it
 * with ctypes and mirrors the librosa keyword signature on top.
 *
 * Conventions
 *   - every function returns 0 on success or a negative MFCC_E_* code; no exception crosses
 *     the ABI; mfcc_last_error() returns a thread-local message for the last failure;
 *   - the caller owns every device buffer; the library allocates device memory only inside
 *     mfcc_plan_create (constant tables) and frees it in mfcc_plan_destroy;
 *   - a plan is immutable after creation and may be shared by several streams, each with its
 *     own workspace;
 *   - "d_" parameters are device pointers, "h_" parameters are host pointers;
 *   - audio is float32 PCM; utterance u occupies d_pcm[sample_offsets[u] .. sample_offsets[u+1]);
 *     its features occupy rows frame_offsets[u] .. frame_offsets[u+1] of the (total_frames,
 *     n_mfcc) row-major float32 output -- i.e. each utterance's slice is exactly the
 *     C-contiguous (n_frames, n_mfcc) array that preprocess.py:135-139 writes to HDF5.
 */
#ifndef MFCC_HIP_H
#define MFCC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFCC_ABI_VERSION 1

enum {
    MFCC_OK = 0,
    MFCC_E_INVALID = -1,      /* bad argument (the librosa ParameterError cases) */
    MFCC_E_UNSUPPORTED = -2,  /* valid librosa configuration this build has no kernel for */
    MFCC_E_HIP = -3,          /* a HIP runtime call failed (no device, launch failure, ...) */
    MFCC_E_WORKSPACE = -4,    /* workspace too small */
    MFCC_E_NOMEM = -5
};

enum { MFCC_PAD_CONSTANT = 0, MFCC_PAD_REFLECT = 1 };
enum { MFCC_MEL_NORM_NONE = 0, MFCC_MEL_NORM_SLANEY = 1 };
enum { MFCC_DCT_NORM_NONE = 0, MFCC_DCT_NORM_ORTHO = 1 };

/* Mirrors the keyword arguments of librosa.feature.mfcc -> melspectrogram -> stft ->
 * filters.mel -> power_to_db (librosa 0.10.1).  mfcc_config_default() fills in librosa's
 * defaults; the reference overrides n_fft=2048, hop_length=512, n_mfcc=13
 * (/root/reference/src/preprocess.py:60-62) and leaves sr at librosa's 22050. */
typedef struct mfcc_config {
    int32_t struct_size;  /* = sizeof(mfcc_config); checked by mfcc_plan_create */
    /* stft */
    int32_t n_fft;        /* 2048 */
    int32_t hop_length;   /* 512 */
    int32_t win_length;   /* 0 -> n_fft */
    int32_t center;       /* 1 */
    int32_t pad_mode;     /* MFCC_PAD_CONSTANT (librosa 0.10 default) or MFCC_PAD_REFLECT */
    float power;          /* 2.0: exponent applied to |STFT| */
    /* filters.mel */
    double sr;            /* 22050 */
    int32_t n_mels;       /* 128 */
    double fmin;          /* 0 */
    double fmax;          /* <= 0 -> sr / 2 */
    int32_t htk;          /* 0 */
    int32_t mel_norm;     /* MFCC_MEL_NORM_SLANEY */
    /* power_to_db */
    float ref;            /* 1.0 */
    float amin;           /* 1e-10 */
    float top_db;         /* 80; negative -> None (no clamp) */
    /* mfcc */
    int32_t n_mfcc;       /* 20 */
    int32_t dct_type;     /* 2 (1 and 3 also supported) */
    int32_t dct_norm;     /* MFCC_DCT_NORM_ORTHO */
    float lifter;         /* 0 */
    /* optional overrides (host pointers, copied by mfcc_plan_create; NULL -> built-in) */
    const double* window;    /* win_length float64 samples; NULL -> periodic Hann */
    const float* mel_basis;  /* (n_mels, 1 + n_fft/2) row-major float32 */
} mfcc_config;

typedef struct mfcc_plan mfcc_plan;

/* librosa defaults (n_mfcc = 20 etc.). */
void mfcc_config_default(mfcc_config* cfg);

/* ABI version, for the loader to check against MFCC_ABI_VERSION. */
int mfcc_abi_version(void);

/* Message for the calling thread's last failure ("" if none). */
const char* mfcc_last_error(void);

/* ---- host-only helpers (no GPU needed): the tables a plan uploads ---------------------- */
/* n_fft float64 values: get_window(window, win_length, fftbins=True) zero-padded (centred) to
 * n_fft, as librosa.stft builds it. */
int mfcc_host_window(const mfcc_config* cfg, double* h_out);
/* (n_mels, 1 + n_fft/2) float32: librosa.filters.mel. */
int mfcc_host_mel_basis(const mfcc_config* cfg, float* h_out);
/* (n_mfcc, n_mels) float32: rows of scipy.fftpack.dct(type, norm) applied along the mel axis. */
int mfcc_host_dct_basis(const mfcc_config* cfg, float* h_out);
/* Frames librosa produces for an utterance of n_samples (1 + n_samples // hop for centred,
 * even n_fft); negative on error (center = 0 and n_samples < n_fft). */
int64_t mfcc_num_frames(const mfcc_config* cfg, int64_t n_samples);
/* frame_offsets[0] = 0, frame_offsets[u+1] = frame_offsets[u] + mfcc_num_frames(len_u). */
int mfcc_batch_layout(const mfcc_config* cfg, const int64_t* h_sample_offsets, int32_t n_utts,
                      int64_t* h_frame_offsets);

/* ---- plan ------------------------------------------------------------------------------- */
/* Builds and uploads window / twiddle / sparse-mel / DCT tables for the current HIP device.
 * Replaces the per-call table construction inside librosa.feature.mfcc
 * (/root/reference/src/preprocess.py:134 rebuilds the filterbank for every utterance). */
int mfcc_plan_create(const mfcc_config* cfg, mfcc_plan** out_plan);
void mfcc_plan_destroy(mfcc_plan* plan);

/* Bytes of device workspace mfcc_execute needs for a batch of this size. */
size_t mfcc_workspace_bytes(const mfcc_plan* plan, int64_t total_frames, int32_t n_utts);

/* ---- execution -------------------------------------------------------------------------- */
/* The whole pipeline of librosa.feature.mfcc for a ragged batch, asynchronous on `stream`
 * (a hipStream_t, or NULL for the default stream):
 *     pad -> frame -> window -> rFFT -> |.|^power -> mel filterbank -> power_to_db (top_db
 *     against the maximum of each utterance's whole mel spectrogram) -> DCT -> [:n_mfcc] -> .T
 * d_out            (total_frames, n_mfcc) float32, row-major.
 * d_group          optional (n_utts) int32: utterances with the same value share one top_db
 *                  maximum (the channels of one multi-channel librosa call); NULL -> every
 *                  utterance is its own group.  Values must be in [0, n_utts); an utterance whose
 *                  value is not gets a maximum of its own and mfcc_workspace_status reports it.
 * d_melspec_out    optional (total_frames, n_mels) float32: receives the mel power
 *                  spectrogram (librosa.feature.melspectrogram, transposed); NULL -> the
 *                  workspace holds it.
 * Replaces: the loop of /root/reference/src/preprocess.py:146-167 calling :131-136. */
int mfcc_execute(const mfcc_plan* plan, const float* d_pcm, const int64_t* d_sample_offsets,
                 const int64_t* d_frame_offsets, const int32_t* d_group, int32_t n_utts,
                 int64_t total_frames, float* d_out, float* d_melspec_out, void* d_workspace,
                 size_t workspace_bytes, void* stream);

/* Same as mfcc_execute with the audio as int16 PCM: sample value = d_pcm16[i] / 32768, exactly
 * the float32 waveform torchaudio decodes a LibriSpeech FLAC into
 * (/root/reference/src/preprocess.py:75 builds the dataset, :132-134 passes its float waveform
 * on), so results are bit-identical to mfcc_execute on that waveform while the upload and the
 * HBM reads are half the size. */
int mfcc_execute_pcm16(const mfcc_plan* plan, const int16_t* d_pcm16, const int64_t* d_sample_offsets,
                       const int64_t* d_frame_offsets, const int32_t* d_group, int32_t n_utts,
                       int64_t total_frames, float* d_out, float* d_melspec_out, void* d_workspace,
                       size_t workspace_bytes, void* stream);

/* mfcc_execute for callers that cannot allocate device memory or hand over a workspace (code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the
 * reference.  Same inputs, outputs and results as mfcc_execute(..., d_melspec_out = NULL, ...); the
 * workspace (tile table, per-utterance maxima, mel power) lives in a static device array of
 * MFCC_NOALLOC_SCRATCH_BYTES inside the library, one copy per GPU, held until the process ends.
 * The call allocates and frees nothing.  A batch whose mfcc_workspace_bytes() exceeds that size is
 * refused with MFCC_E_WORKSPACE (at the reference's settings about 130 000 frames, 1.1 hours of
 * 16 kHz audio).  The plan's device must be current.  All calls on one GPU share that GPU's array:
 * calls that overlap in time on one GPU (different streams or threads) race; calls on one stream,
 * or otherwise ordered, are safe -- every call initialises what it reads.  The scratch is not
 * exposed, so mfcc_workspace_status cannot be asked about it: a caller of this entry point vouches
 * for d_group itself (NULL, or values in [0, n_utts); a value outside still cannot index out of
 * bounds, its utterance just gets a maximum of its own). */
#define MFCC_NOALLOC_SCRATCH_BYTES (64u * 1024u * 1024u)
int mfcc_execute_noalloc(const mfcc_plan* plan, const float* d_pcm, const int64_t* d_sample_offsets,
                         const int64_t* d_frame_offsets, const int32_t* d_group, int32_t n_utts,
                         int64_t total_frames, float* d_out, void* stream);

/* Waits for `stream` and reports what the kernels of the last mfcc_execute* on this workspace
 * found wrong with their device-side inputs: MFCC_E_INVALID if a d_group value lay outside
 * [0, n_utts) (the kernels give such an utterance a maximum of its own instead of indexing out of
 * bounds), else MFCC_OK.  Optional: for callers that cannot vouch for d_group.  No counterpart in
 * the reference. */
int mfcc_workspace_status(const void* d_workspace, void* stream);

/* |STFT|^power only: d_spec_out is (total_frames, 1 + n_fft/2) float32 row-major, i.e.
 * np.abs(librosa.stft(y))**power transposed (/root/reference/src/preprocess.py:111-113). */
int mfcc_execute_spectrogram(const mfcc_plan* plan, const float* d_pcm,
                             const int64_t* d_sample_offsets, const int64_t* d_frame_offsets,
                             int32_t n_utts, int64_t total_frames, float* d_spec_out,
                             void* d_workspace, size_t workspace_bytes, void* stream);

/* ---- measurement ------------------------------------------------------------------------- */
/* With profiling enabled, mfcc_execute records HIP events on its stream around the front-end
 * kernel (pad/frame/window/FFT/mel) and the dB+DCT kernel; mfcc_plan_last_timings waits for the
 * last call and returns both durations in milliseconds.  One execute in flight per plan while
 * profiling.  No counterpart in the reference (it has no timing code). */
int mfcc_plan_set_profiling(mfcc_plan* plan, int enable);
int mfcc_plan_last_timings(const mfcc_plan* plan, float* ms_front, float* ms_db_dct);
/* The plan keeps the events of its last 256 executes since profiling was last enabled (enabling it
 * again forgets them).  Copies the durations of the most recent min(max_n, recorded, 256) executes,
 * oldest first, into ms_front / ms_db_dct (either may be NULL) after waiting for the last of them;
 * returns how many, or a negative MFCC_E_* code.  Lets a timing loop run its executes back to back
 * and read the per-kernel times afterwards.  No counterpart in the reference. */
int mfcc_plan_timing_history(const mfcc_plan* plan, int32_t max_n, float* ms_front, float* ms_db_dct);
/* Name of the front-end kernel the plan runs: "fft2048" (the reference's n_fft,
 * /root/reference/src/preprocess.py:134), "fft256", "fft320", "fft400", "fft480", "fft512", "fft640", "fft800", "fft960", "fft1024" (in-register FFTs for the
 * short transforms of 16 kHz front ends) or "dense_dft" (any other even n_fft: the DFT as an fp32 MFMA
 * contraction).  A static string; NULL for a NULL plan.  No counterpart in the reference. */
const char* mfcc_plan_front_kernel(const mfcc_plan* plan);

/* ---- global CMVN (extension; the reference does not normalise) -------------------------- */
/* Adds, per coefficient, sum(x) and sum(x^2) over the rows of d_feats plus the row count
 * into d_stats[0..n_mfcc), d_stats[n_mfcc..2 n_mfcc), d_stats[2 n_mfcc] (float64, caller
 * zeroes them).  The caller all-reduces d_stats across GPUs (RCCL sum) and then applies. */
int mfcc_cmvn_accumulate(const float* d_feats, int64_t n_rows, int32_t n_mfcc, double* d_stats,
                         void* stream);
/* In place: x = (x - mean) / sqrt(max(var, eps)) with mean/var from d_stats. */
int mfcc_cmvn_apply(float* d_feats, int64_t n_rows, int32_t n_mfcc, const double* d_stats,
                    double eps, void* stream);

/* ---- dB scaling of a spectrogram (the reference's display path) --------------------------- */
/* librosa.power_to_db (amplitude = 0) or librosa.amplitude_to_db (amplitude = 1) of the n floats at
 * d_in: 10 log10(max(amin, v)) - 10 log10(max(amin, ref)), clamped at (its maximum over the whole
 * array - top_db) when top_db >= 0; v = |x| or, for amplitudes, x^2 with ref and amin squared.
 * ref_is_max != 0 takes the array's maximum as the reference (librosa's ref=np.max).  d_out may
 * equal d_in; d_workspace is 256 bytes of device memory.  Replaces
 * librosa.amplitude_to_db(np.abs(stft)) at /root/reference/src/preprocess.py:113-114. */
int mfcc_to_db(const float* d_in, int64_t n, int32_t amplitude, float ref, int32_t ref_is_max, float amin,
               float top_db, float* d_out, void* d_workspace, void* stream);

/* ---- MFCC of a log-mel spectrogram that already exists ------------------------------------- */
/* The S= form of librosa.feature.mfcc (the reference's call at /root/reference/src/preprocess.py:134
 * passes y; librosa's other spelling of the same function takes the dB-scaled mel spectrogram):
 * scipy.fftpack.dct of the plan's type and norm along the mel axis, the first n_mfcc coefficients,
 * the plan's lifter.  d_logmel is (rows, n_mels) row-major -- a frame per row, as mfcc_execute's
 * output is (rows, n_mfcc) -- and n_mels is the plan's.  Asynchronous on `stream`. */
int mfcc_dct(const mfcc_plan* plan, const float* d_logmel, int64_t rows, float* d_out, void* stream);

/* ---- delta features (extension; the reference trains on the plain MFCCs) ------------------ */
/* librosa.feature.delta(data, width, order, axis=time, mode='interp') on the concatenated
 * (total_frames, n_coef) features of a ragged batch, every utterance on its own: the order-th
 * Savitzky-Golay derivative of a degree-`order` fit over `width` frames; frames within width/2 of
 * an utterance's ends take the derivative of the polynomial fitted to its first / last `width`
 * frames.  h_frame_offsets is the host copy of d_frame_offsets (the width check needs the lengths:
 * an utterance shorter than `width` is an error, as in librosa).  d_out must not alias d_feats. */
int mfcc_delta(const float* d_feats, const int64_t* d_frame_offsets, const int64_t* h_frame_offsets, int32_t n_utts,
               int32_t n_coef, int32_t width, int32_t order, float* d_out, void* stream);

/* ---- host helper for the 16-bit ingest ----------------------------------------------------- */
/* h_dst[i] = h_src[i] * 32768 as int16.  Returns 1 if every sample lies on the int16 grid (the
 * product is an integer in [-32768, 32767]: what torchaudio yields for the 16-bit FLAC of
 * /root/reference/src/preprocess.py:75), 0 if not (then h_dst is not to be used), negative on bad
 * arguments.  Host memory only, no GPU work; thread-safe. */
int mfcc_host_pcm16_from_float(const float* h_src, int64_t n, int16_t* h_dst);

/* ---- host helper: the per-utterance feature file ------------------------------------------ */
/* Writes a new HDF5 file holding one contiguous float32 dataset `dataset_name` of shape
 * (rows, cols) with n_attrs scalar attributes: attribute i is named attr_names[i] and is the
 * int64 attr_ints[i], or, where attr_is_string[i] != 0, the variable-length UTF-8 string
 * attr_strings[i].  With dataset "mfccs" and the attributes speaker_id, chapter_id, utterance_id,
 * sample_rate (ints) and transcript (string) this is the file of
 * /root/reference/src/preprocess.py:138-144, readable by h5py as /root/reference/src/trainer.py:110-114
 * reads it.  Host memory and file I/O only; thread-safe.  MFCC_OK or MFCC_E_INVALID
 * (mfcc_host_file_error() tells why). */
int mfcc_host_write_feature_file(const char* path, const char* dataset_name, const float* h_data, int64_t rows,
                                 int64_t cols, int32_t n_attrs, const char* const* attr_names,
                                 const int32_t* attr_is_string, const int64_t* attr_ints,
                                 const char* const* attr_strings);
/* The same for n_files files in one call: file i gets h_data[i] (rows[i], cols) and the attribute
 * values attr_ints / attr_strings [i * n_attrs .. i * n_attrs + n_attrs); names and kinds are shared.
 * Stops at the first failure. */
int mfcc_host_write_feature_files(int32_t n_files, const char* const* paths, const char* dataset_name,
                                  const float* const* h_data, const int64_t* rows, int64_t cols, int32_t n_attrs,
                                  const char* const* attr_names, const int32_t* attr_is_string,
                                  const int64_t* attr_ints, const char* const* attr_strings);
/* The same with the files dealt to n_threads host threads (file creation is system-call bound;
 * n_threads <= 1 writes them in order on the caller's thread).  The drop-in step writes one file
 * per utterance (/root/reference/src/preprocess.py:138-144 inside the loop at :155-167): after
 * the GPU path that is where its wall clock goes. */
int mfcc_host_write_feature_files_mt(int32_t n_files, const char* const* paths, const char* dataset_name,
                                     const float* const* h_data, const int64_t* rows, int64_t cols, int32_t n_attrs,
                                     const char* const* attr_names, const int32_t* attr_is_string,
                                     const int64_t* attr_ints, const char* const* attr_strings, int32_t n_threads);
const char* mfcc_host_file_error(void);

/* ---- host helper: packing a ragged batch for upload ----------------------------------------- */
/* Copies the samples [a, b) of the concatenation of n_utts host utterances -- srcs[u] holds
 * offsets[u + 1] - offsets[u] elements, offsets[0] = 0 -- to dst[a - c0 ..], dst being a staging
 * buffer (pinned host memory) that starts at sample c0 of the concatenation.  mode 0: float32
 * samples copied; 1: int16 PCM copied; 2: float32 samples on the int16 grid converted to int16 as
 * mfcc_host_pcm16_from_float does.  Returns 1, or 0 when mode 2 meets a sample off the grid (dst is
 * then not to be used), negative on bad arguments.  Replaces the per-utterance hand-over of
 * /root/reference/src/preprocess.py:132-134 (one host array per librosa call) by one device
 * buffer per batch.  Host memory only; thread-safe: callers split [c0, c1) over their threads. */
int mfcc_host_pack_range(const void* const* srcs, const int64_t* offsets, int64_t n_utts, void* dst, int64_t c0,
                         int64_t a, int64_t b, int32_t mode);

#ifdef __cplusplus
}
#endif
#endif /* MFCC_HIP_H */
