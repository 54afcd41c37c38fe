// Native writer of the per-utterance feature file: one contiguous float32 dataset with scalar
// attributes (int64, or variable-length UTF-8 strings kept in a global heap collection) in a
// minimal HDF5 file -- superblock v0, root group with a v1 B-tree / local heap / one symbol-table
// node, v1 object headers.  Byte for byte the file speech_to_text_amd/hdf5_min.py writes (the tests
// compare them); that layout is checked against libhdf5's h5dump.  The schema is the reference's:
// dataset `mfccs` (F, 13) plus five attributes, /root/reference/src/preprocess.py:138-144.
// Host code only.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <sys/uio.h>
#include <unistd.h>

#include "../../include/mfcc_hip.h"

namespace {

typedef std::string Bytes;
const uint64_t UNDEF = 0xFFFFFFFFFFFFFFFFull;

template <typename T>
void put(Bytes& b, T v) { b.append(reinterpret_cast<const char*>(&v), sizeof(T)); }   // little-endian host
void zeros(Bytes& b, size_t n) { b.append(n, '\0'); }
void pad8(Bytes& b) { zeros(b, (8 - b.size() % 8) % 8); }
Bytes padded(Bytes b) { pad8(b); return b; }

Bytes dt_f32() {
    Bytes b;
    put<uint8_t>(b, 0x11); put<uint8_t>(b, 0x20); put<uint8_t>(b, 31); put<uint8_t>(b, 0); put<uint32_t>(b, 4);
    put<uint16_t>(b, 0); put<uint16_t>(b, 32); put<uint8_t>(b, 23); put<uint8_t>(b, 8); put<uint8_t>(b, 0);
    put<uint8_t>(b, 23); put<uint32_t>(b, 127);
    return b;
}
Bytes dt_i64() {
    Bytes b;
    put<uint8_t>(b, 0x10); put<uint8_t>(b, 0x08); put<uint8_t>(b, 0); put<uint8_t>(b, 0); put<uint32_t>(b, 8);
    put<uint16_t>(b, 0); put<uint16_t>(b, 64);
    return b;
}
Bytes dt_vlen_utf8() {
    Bytes b;
    put<uint8_t>(b, 0x19); put<uint8_t>(b, 0x01); put<uint8_t>(b, 0x01); put<uint8_t>(b, 0); put<uint32_t>(b, 16);
    put<uint8_t>(b, 0x10); put<uint8_t>(b, 0x00); put<uint8_t>(b, 0); put<uint8_t>(b, 0); put<uint32_t>(b, 1);
    put<uint16_t>(b, 0); put<uint16_t>(b, 8);
    return b;
}
Bytes ds_simple(const std::vector<uint64_t>& shape) {
    Bytes b;
    put<uint8_t>(b, 1); put<uint8_t>(b, (uint8_t)shape.size()); put<uint8_t>(b, 0); zeros(b, 5);
    for (uint64_t d : shape) put<uint64_t>(b, d);
    return b;
}
Bytes message(uint16_t type, const Bytes& data, uint8_t flags = 0) {
    const Bytes d = padded(data);
    Bytes b;
    put<uint16_t>(b, type); put<uint16_t>(b, (uint16_t)d.size()); put<uint8_t>(b, flags); zeros(b, 3);
    return b + d;
}
Bytes attribute(const std::string& name, const Bytes& dt, const Bytes& ds, const Bytes& data) {
    Bytes nm = name; nm.push_back('\0');
    Bytes body;
    put<uint8_t>(body, 1); zeros(body, 1);
    put<uint16_t>(body, (uint16_t)nm.size()); put<uint16_t>(body, (uint16_t)dt.size()); put<uint16_t>(body, (uint16_t)ds.size());
    body += padded(nm) + padded(dt) + padded(ds) + data;
    return message(0x000C, body);
}
Bytes object_header(const Bytes& messages, uint16_t n) {
    Bytes b;
    put<uint8_t>(b, 1); zeros(b, 1); put<uint16_t>(b, n); put<uint32_t>(b, 1); put<uint32_t>(b, (uint32_t)messages.size());
    zeros(b, 4);
    return b + messages;
}

thread_local std::string g_file_err;

// "<directory of path>.partial", or "" for a path without a directory part
std::string side_dir_of(const std::string& full) {
    const size_t slash = full.find_last_of('/');
    if (slash == std::string::npos || slash == 0) return std::string();
    std::string dir = full.substr(0, slash);
    while (dir.size() > 1 && dir.back() == '/') dir.pop_back();
    return dir + ".partial";
}

// All of `iov` to fd (writev may stop short).
bool write_all(int fd, struct iovec* iov, int n) {
    while (n > 0) {
        ssize_t w = ::writev(fd, iov, n);
        if (w < 0) return false;
        while (n > 0 && (size_t)w >= iov->iov_len) { w -= (ssize_t)iov->iov_len; ++iov; --n; }
        if (n > 0) { iov->iov_base = static_cast<char*>(iov->iov_base) + w; iov->iov_len -= (size_t)w; }
    }
    return true;
}

// side_ready: the caller has already created the side directory of the temporaries (batch writes
// create it once instead of once per file).
int write_one(const char* path, const char* dataset_name, const float* h_data, int64_t rows,
              int64_t cols, int32_t n_attrs, const char* const* attr_names,
              const int32_t* attr_is_string, const int64_t* attr_ints,
              const char* const* attr_strings, bool side_ready) {
    if (!path || !dataset_name || rows < 0 || cols < 0 || n_attrs < 0 || (rows * cols > 0 && !h_data) ||
        (n_attrs > 0 && (!attr_names || !attr_is_string || !attr_ints || !attr_strings))) {
        g_file_err = "bad arguments";
        return MFCC_E_INVALID;
    }
    const uint64_t nbytes = (uint64_t)rows * (uint64_t)cols * 4;
    // global heap collection with the string attributes
    std::vector<Bytes> strs;
    std::vector<int> str_index(n_attrs, 0);
    uint64_t heap_size = 16;
    for (int i = 0; i < n_attrs; ++i)
        if (attr_is_string[i]) {
            strs.emplace_back(attr_strings[i] ? attr_strings[i] : "");
            str_index[i] = (int)strs.size();
            heap_size += 16 + (strs.back().size() + 7) / 8 * 8;
        }
    const uint64_t gcol_bytes = strs.empty() ? 0 : std::max<uint64_t>(4096, (heap_size + 16 + 7) / 8 * 8);

    // dataset object header, addresses patched in below
    Bytes msgs = message(0x0001, ds_simple({(uint64_t)rows, (uint64_t)cols})) + message(0x0003, dt_f32(), 1);
    {
        Bytes fill;
        put<uint8_t>(fill, 2); put<uint8_t>(fill, 2); put<uint8_t>(fill, 2); put<uint8_t>(fill, 1); put<uint32_t>(fill, 0);
        msgs += message(0x0005, fill);
    }
    const size_t data_at = 16 + msgs.size() + 8 + 2;
    {
        Bytes layout;
        put<uint8_t>(layout, 3); put<uint8_t>(layout, 1); put<uint64_t>(layout, 0); put<uint64_t>(layout, nbytes);
        msgs += message(0x0008, layout);
    }
    std::vector<size_t> heap_at;
    for (int i = 0; i < n_attrs; ++i) {
        Bytes m;
        if (attr_is_string[i]) {
            Bytes ref;
            put<uint32_t>(ref, (uint32_t)strs[str_index[i] - 1].size()); put<uint64_t>(ref, 0);
            put<uint32_t>(ref, (uint32_t)str_index[i]);
            m = attribute(attr_names[i], dt_vlen_utf8(), ds_simple({}), ref);
            heap_at.push_back(16 + msgs.size() + m.size() - 16 + 4);
        } else {
            Bytes v;
            put<int64_t>(v, attr_ints[i]);
            m = attribute(attr_names[i], dt_i64(), ds_simple({}), v);
        }
        msgs += m;
    }
    Bytes header = object_header(msgs, (uint16_t)(4 + n_attrs));

    // fixed-size pieces and addresses
    const uint64_t SB = 96, ROOT_OH = 40, BTREE = 544, HEAP_HDR = 32, SNOD = 328;
    Bytes heap_data(8, '\0');
    const uint64_t name_off = heap_data.size();
    {
        Bytes nm = dataset_name; nm.push_back('\0');
        heap_data += padded(nm);
    }
    put<uint64_t>(heap_data, 1); put<uint64_t>(heap_data, 16);
    const uint64_t a_root = SB, a_btree = a_root + ROOT_OH, a_heap = a_btree + BTREE, a_heap_data = a_heap + HEAP_HDR;
    const uint64_t a_snod = a_heap_data + heap_data.size(), a_dset = a_snod + SNOD;
    const uint64_t a_gcol = a_dset + header.size(), a_data = a_gcol + gcol_bytes;
    const uint64_t eof = a_data + (nbytes + 7) / 8 * 8;
    const uint64_t data_addr = nbytes ? a_data : UNDEF;
    std::memcpy(&header[data_at], &data_addr, 8);
    for (size_t at : heap_at) std::memcpy(&header[at], &a_gcol, 8);

    Bytes f;
    f.reserve((size_t)(a_data + 64));
    f.append("\x89HDF\r\n\x1a\n", 8);
    for (uint8_t v : {0, 0, 0, 0, 0, 8, 8, 0}) put<uint8_t>(f, v);
    put<uint16_t>(f, 4); put<uint16_t>(f, 16); put<uint32_t>(f, 0);
    put<uint64_t>(f, 0); put<uint64_t>(f, UNDEF); put<uint64_t>(f, eof); put<uint64_t>(f, UNDEF);
    put<uint64_t>(f, 0); put<uint64_t>(f, a_root); put<uint32_t>(f, 1); zeros(f, 4); put<uint64_t>(f, a_btree);
    put<uint64_t>(f, a_heap);
    {
        Bytes st;
        put<uint64_t>(st, a_btree); put<uint64_t>(st, a_heap);
        f += object_header(message(0x0011, st), 1);
    }
    {
        Bytes bt("TREE");
        put<uint8_t>(bt, 0); put<uint8_t>(bt, 0); put<uint16_t>(bt, 1); put<uint64_t>(bt, UNDEF); put<uint64_t>(bt, UNDEF);
        put<uint64_t>(bt, 0); put<uint64_t>(bt, a_snod); put<uint64_t>(bt, name_off);
        zeros(bt, BTREE - bt.size());
        f += bt;
    }
    f.append("HEAP", 4);
    put<uint8_t>(f, 0); zeros(f, 3); put<uint64_t>(f, heap_data.size()); put<uint64_t>(f, heap_data.size() - 16);
    put<uint64_t>(f, a_heap_data);
    f += heap_data;
    {
        Bytes sn("SNOD");
        put<uint8_t>(sn, 1); zeros(sn, 1); put<uint16_t>(sn, 1);
        put<uint64_t>(sn, name_off); put<uint64_t>(sn, a_dset); put<uint32_t>(sn, 0); zeros(sn, 4 + 16);
        zeros(sn, SNOD - sn.size());
        f += sn;
    }
    f += header;
    if (!strs.empty()) {
        Bytes body;
        for (size_t i = 0; i < strs.size(); ++i) {
            put<uint16_t>(body, (uint16_t)(i + 1)); put<uint16_t>(body, 1); zeros(body, 4); put<uint64_t>(body, strs[i].size());
            body += padded(strs[i]);
        }
        const uint64_t free_bytes = gcol_bytes - 16 - body.size();
        f.append("GCOL", 4);
        put<uint8_t>(f, 1); zeros(f, 3); put<uint64_t>(f, gcol_bytes);
        f += body;
        put<uint16_t>(f, 0); put<uint16_t>(f, 0); zeros(f, 4); put<uint64_t>(f, free_bytes);
        zeros(f, free_bytes - 16);
    }
    if (f.size() != a_data) {
        g_file_err = "internal layout error";
        return MFCC_E_INVALID;
    }

    // Written under a temporary name and renamed once complete.  The reference's reader opens every
    // file it finds under the feature directory, whatever its name
    // (/root/reference/src/trainer.py:106-110), so the temporary lives OUTSIDE that directory: in a
    // sibling "<directory>.partial" on the same file system (rename stays atomic).  A killed writer
    // leaves its half-written file there, where no reader walks.  A path without a directory part
    // has no sibling to use and takes "<path>.tmp".
    const std::string full(path);
    const std::string side = side_dir_of(full);
    std::string tmp;
    if (side.empty()) {
        tmp = full + ".tmp";
    } else {
        if (!side_ready) (void)::mkdir(side.c_str(), 0777);
        tmp = side + "/" + full.substr(full.find_last_of('/') + 1) + "." + std::to_string((long)::getpid());
    }
    int fd = ::open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
    // Another writer (a thread of this batch, or another rank that shares the directory) may have
    // removed the side directory while it was empty: make it again and retry, a few times in case
    // the two keep crossing.
    for (int attempt = 0; fd < 0 && !side.empty() && errno == ENOENT && attempt < 16; ++attempt) {
        (void)::mkdir(side.c_str(), 0777);
        fd = ::open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
    }
    if (fd < 0) {
        g_file_err = std::string("cannot open ") + tmp;
        return MFCC_E_INVALID;
    }
    // metadata, dataset and padding leave in one system call
    static const char pad[8] = {0};
    struct iovec iov[3];
    int n_iov = 0;
    iov[n_iov].iov_base = const_cast<char*>(f.data()); iov[n_iov].iov_len = f.size(); ++n_iov;
    if (nbytes) { iov[n_iov].iov_base = const_cast<float*>(h_data); iov[n_iov].iov_len = (size_t)nbytes; ++n_iov; }
    if (nbytes % 8) { iov[n_iov].iov_base = const_cast<char*>(pad); iov[n_iov].iov_len = (size_t)(8 - nbytes % 8); ++n_iov; }
    bool ok = write_all(fd, iov, n_iov);
    ok = (::close(fd) == 0) && ok;
    if (!ok) {
        (void)::unlink(tmp.c_str());
        g_file_err = std::string("short write to ") + tmp;
        return MFCC_E_INVALID;
    }
    if (::rename(tmp.c_str(), path) != 0) {
        (void)::unlink(tmp.c_str());
        g_file_err = std::string("cannot rename ") + tmp + " to " + path;
        return MFCC_E_INVALID;
    }
    return MFCC_OK;
}

}  // namespace

// No C++ exception may cross the C ABI.
#define MFCC_FILE_CATCH_ALL                                                  \
    catch (const std::bad_alloc&) {                                          \
        try { g_file_err = "out of host memory"; } catch (...) {}            \
        return MFCC_E_NOMEM;                                                 \
    } catch (...) {                                                          \
        try { g_file_err = "unexpected C++ exception"; } catch (...) {}      \
        return MFCC_E_INVALID;                                               \
    }

extern "C" {

const char* mfcc_host_file_error(void) { return g_file_err.c_str(); }

int mfcc_host_write_feature_file(const char* path, const char* dataset_name, const float* h_data, int64_t rows,
                                 int64_t cols, int32_t n_attrs, const char* const* attr_names,
                                 const int32_t* attr_is_string, const int64_t* attr_ints,
                                 const char* const* attr_strings) try {
    return write_one(path, dataset_name, h_data, rows, cols, n_attrs, attr_names, attr_is_string, attr_ints,
                     attr_strings, false);
} MFCC_FILE_CATCH_ALL

int mfcc_host_write_feature_files(int32_t n_files, const char* const* paths, const char* dataset_name,
                                  const float* const* h_data, const int64_t* rows, int64_t cols, int32_t n_attrs,
                                  const char* const* attr_names, const int32_t* attr_is_string,
                                  const int64_t* attr_ints, const char* const* attr_strings) try {
    return mfcc_host_write_feature_files_mt(n_files, paths, dataset_name, h_data, rows, cols, n_attrs, attr_names,
                                            attr_is_string, attr_ints, attr_strings, 1);
} MFCC_FILE_CATCH_ALL

int mfcc_host_write_feature_files_mt(int32_t n_files, const char* const* paths, const char* dataset_name,
                                     const float* const* h_data, const int64_t* rows, int64_t cols, int32_t n_attrs,
                                     const char* const* attr_names, const int32_t* attr_is_string,
                                     const int64_t* attr_ints, const char* const* attr_strings, int32_t n_threads) try {
    if (n_files < 0 || (n_files > 0 && (!paths || !h_data || !rows))) {
        g_file_err = "bad arguments";
        return MFCC_E_INVALID;
    }
    if (n_files == 0) return MFCC_OK;
    for (int32_t i = 0; i < n_files; ++i)
        if (!paths[i]) {
            g_file_err = "bad arguments";
            return MFCC_E_INVALID;
        }
    // the side directories of the temporaries: created once per distinct directory, not per file
    std::vector<std::string> sides;
    for (int32_t i = 0; i < n_files; ++i) {
        const std::string side = side_dir_of(paths[i]);
        if (!side.empty() && (sides.empty() || sides.back() != side)) {
            bool seen = false;
            for (const std::string& s : sides) seen = seen || s == side;
            if (!seen) {
                (void)::mkdir(side.c_str(), 0777);
                sides.push_back(side);
            }
        }
    }
    // files are handed out one at a time from a shared counter; the first failure stops the hand-out
    std::atomic<int32_t> next(0);
    std::atomic<int> status(MFCC_OK);
    std::string first_err;
    std::atomic_flag err_lock = ATOMIC_FLAG_INIT;
    auto work = [&]() {
        for (;;) {
            const int32_t i = next.fetch_add(1);
            if (i >= n_files || status.load() != MFCC_OK) return;
            const int rc = write_one(paths[i], dataset_name, h_data[i], rows[i], cols, n_attrs, attr_names, attr_is_string,
                                     attr_ints + (size_t)i * n_attrs, attr_strings + (size_t)i * n_attrs, true);
            if (rc != MFCC_OK) {
                while (err_lock.test_and_set()) {}
                if (status.load() == MFCC_OK) { first_err = g_file_err; status.store(rc); }
                err_lock.clear();
                return;
            }
        }
    };
    const int T = std::max(1, std::min<int>({n_threads, n_files, 64}));
    if (T == 1) {
        work();
    } else {
        std::vector<std::thread> pool;
        pool.reserve(T - 1);
        for (int t = 1; t < T; ++t) {
            try {
                pool.emplace_back(work);
            } catch (...) {          // no more threads to be had: the ones running share the files
                break;
            }
        }
        work();
        for (std::thread& th : pool) th.join();
    }
    for (const std::string& s : sides) (void)::rmdir(s.c_str());   // succeeds only if this batch left it empty
    if (status.load() != MFCC_OK) {
        g_file_err = first_err;      // (the failing thread's message, for the caller's thread)
        return status.load();
    }
    return MFCC_OK;
} MFCC_FILE_CATCH_ALL

/* Packs the samples [a, b) of the concatenation of n_utts host utterances (srcs[u] holds
 * offsets[u+1] - offsets[u] elements) into dst[a - c0 ...].  mode 0: float32 -> float32,
 * 1: int16 -> int16, 2: float32 on the int16 grid -> int16 (returns 0 if a sample is off the grid). */
int mfcc_host_pack_range(const void* const* srcs, const int64_t* offsets, int64_t n_utts, void* dst, int64_t c0,
                         int64_t a, int64_t b, int32_t mode) {
    if (!srcs || !offsets || !dst || n_utts <= 0 || a < c0 || b < a || mode < 0 || mode > 2) return a == b ? 1 : -1;
    // the utterance that holds sample a
    int64_t lo = 0, hi = n_utts;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) / 2;
        if (offsets[mid] <= a) lo = mid; else hi = mid;
    }
    int64_t u = lo;
    while (a < b) {
        while (u + 1 < n_utts && offsets[u + 1] <= a) ++u;
        const int64_t e = std::min<int64_t>(b, offsets[u + 1]);
        const int64_t n = e - a, at = a - offsets[u];
        if (n <= 0) return -1;
        if (mode == 0) {
            std::memcpy(static_cast<float*>(dst) + (a - c0), static_cast<const float*>(srcs[u]) + at, (size_t)n * 4);
        } else if (mode == 1) {
            std::memcpy(static_cast<int16_t*>(dst) + (a - c0), static_cast<const int16_t*>(srcs[u]) + at, (size_t)n * 2);
        } else {
            if (mfcc_host_pcm16_from_float(static_cast<const float*>(srcs[u]) + at, n, static_cast<int16_t*>(dst) + (a - c0)) != 1)
                return 0;
        }
        a = e;
    }
    return 1;
}

}  // extern "C"
