// Host-side (CPU, double precision) construction of the constant tables a plan uploads:
// the analysis window, the mel filterbank, the DCT basis and their device-friendly packings.
// Restates librosa 0.10.1 filters.get_window / filters.mel / convert.mel_frequencies and
// scipy.fftpack.dct; the call that needs them is /root/reference/src/preprocess.py:134.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/mfcc_hip.h"

namespace mfcc_host {

// Validated, defaults-resolved copy of an mfcc_config (win_length, fmax filled in).
struct Resolved {
    mfcc_config c;
    int n_bins;  // 1 + n_fft / 2
};

// Returns "" on success, else the reason (the librosa ParameterError cases).
std::string resolve(const mfcc_config* in, Resolved* out);

void build_window(const Resolved& r, std::vector<double>& w);        // n_fft values
void build_mel_basis(const Resolved& r, std::vector<float>& m);      // n_mels x n_bins
void build_dct_basis(const Resolved& r, std::vector<float>& d);      // n_mfcc x n_mels
void build_lifter(const Resolved& r, std::vector<float>& l);         // n_mfcc factors

int64_t num_frames(const Resolved& r, int64_t n_samples);

// Sparse packing of the mel filterbank for the lane-per-filter contraction.
// Filters are sorted by support length (longest first) and dealt 64 per pass; in pass p lane l
// owns filter filt[p*64+l] (or -1) and reads bins four at a time: start[p*64+l] is a multiple
// of 4, quad j covers bins start + 4j .. start + 4j + 3 with weights
// w[((round_off[p]/4 + j)*64 + l)*4 + c] (zero outside the filter's support).  rounds[p] is the
// number of bins a lane reads in pass p.  Starts and lanes are chosen so that the 16 lanes the
// hardware services together read from distinct LDS bank groups.
struct SparseMel {
    int n_pass = 0;
    std::vector<int> rounds;     // per pass, a multiple of 4 * chunk_quads
    std::vector<int> round_off;  // per pass, prefix sum of rounds
    std::vector<int32_t> start;  // n_pass * 64
    std::vector<int32_t> filt;   // n_pass * 64
    std::vector<float> w;        // total_rounds * 64
    int total_rounds = 0;
    int max_read = 0;            // largest bin index any lane reads
};
// chunk_quads: rounds[p] / 4 is padded to a multiple of this.
void build_sparse_mel(const std::vector<float>& basis, int n_mels, int n_bins, int chunk_quads, SparseMel& s);

// Load-balanced packing of the mel filterbank for the 2048-point kernel.
// The power spectrum lives in rows of row_bins bins, row r starting at float r * row_stride
// (row_stride - row_bins floats of padding after each row).  Every filter's support is cut at
// the row boundaries and then into jobs of `quads` consecutive 4-bin quads; each lane of the
// wavefront runs n_jobs >= 4 jobs per frame, one per slot.  Consecutive jobs of a filter form a
// chain that sits in consecutive slots of one lane, so the lane adds them up in a register;
// a filter with more jobs than slots has several chains.  Job j of lane l:
//   at[j*64+l]    float index of the job's first quad (a multiple of 4) in the padded layout
//   cont[j*64+l]  1 if the job continues the chain of slot j - 1 (add that slot's running sum)
//   cell[j*64+l]  where the running sum goes after this job, in floats from the start of the
//                 cell area: a real cell for the last job of a chain, else row n_rows, cell l
//   w[((j*quads + i)*64 + l)*4 + c]   weight of bin c of quad i
// Cell area: for each group p of 64 filters, rows[p] rows of 64 cells; chain c of filter
// f = 64 p + x owns cell x of row c of its group, and the filter's output is the sum over its
// group's rows (rows with unwritten cells -- bits of zero_rows -- must be cleared each frame).
// If that takes more rows than allowed (sum of rows[p] <= max_rows, rows[p] <=
// max_rows_per_group) `shared` is set instead: rows[p] = 1 and all chains of a filter add
// into its one cell.
// Chains are dealt to lanes so that the 16 lanes a ds_read_b128 services together read from
// distinct LDS bank groups wherever the filter geometry allows it.
struct MelJobs {
    int n_jobs = 0;
    int quads = 0;
    std::vector<int32_t> at;
    std::vector<int32_t> cell;
    std::vector<int32_t> cont;
    std::vector<float> w;
    int rows[4] = {0, 0, 0, 0};
    int n_rows = 0;         // sum of rows[]
    uint32_t zero_rows = 0;
    bool shared = false;
    int max_read = 0;       // largest float index any job reads
    int bank_conflicts = 0; // jobs that share a bank group with another job of their read group
};
void build_mel_jobs(const std::vector<float>& basis, int n_mels, int n_bins, int quads, int row_bins,
                    int row_stride, int max_rows, int max_rows_per_group, MelJobs& out);

}  // namespace mfcc_host
