// chain_reader.cpp -- libmcechains.so (see include/mcechains.h): mmap + multi-threaded parse of
// whitespace-separated numeric text (CosmoMC `root_N.txt`, MontePython chains).
//
// Replaces np.loadtxt at reference MCEvidence.py:564.  Every field is converted with Clinger's
// exact fast path (<= 2^53 mantissa, |10^e| <= 10^22: one correctly rounded IEEE operation) and
// falls back to strtod (correctly rounded in glibc) otherwise, so the values are bit-identical to
// Python's float() -- which is what np.loadtxt applies.  Host-only C++17; no GPU involved.
#include "../../include/mcechains.h"

#include <fcntl.h>
#include <locale.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

const double kP10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                         1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\v' || c == '\f'; }
inline bool is_digit(char c) { return c >= '0' && c <= '9'; }

locale_t c_locale()
{
    static locale_t loc = newlocale(LC_ALL_MASK, "C", (locale_t)0);
    return loc;
}

// strtod on a copy of the token; accepts what Python's float() accepts for decimal text
bool parse_slow(const char* p, const char* e, double* out)
{
    const size_t n = (size_t)(e - p);
    if (n == 0 || n > 4096) return false;
    for (const char* q = p; q < e; ++q) {
        const char c = *q;
        const bool ok = is_digit(c) || c == '+' || c == '-' || c == '.' || c == 'e' || c == 'E' ||
                        ((c | 0x20) >= 'a' && (c | 0x20) <= 'z' && (c | 0x20) != 'x' && (c | 0x20) != 'p');
        if (!ok) return false;
    }
    char buf[4100];
    std::memcpy(buf, p, n);
    buf[n] = '\0';
    char* endp = nullptr;
    errno = 0;
    const double v = strtod_l(buf, &endp, c_locale());
    if (endp != buf + n) return false;
    *out = v;
    return true;
}

// one numeric token [p, e) -> correctly rounded double
bool parse_token(const char* p, const char* e, double* out)
{
    const char* const tok = p;
    bool neg = false;
    if (p < e && (*p == '+' || *p == '-')) {
        neg = (*p == '-');
        ++p;
    }
    uint64_t mant = 0;
    int shift = 0;           // decimal exponent adjustment from the digits themselves
    bool any = false, inexact = false;
    constexpr uint64_t kMantMax = (UINT64_MAX - 9) / 10;
    while (p < e && is_digit(*p)) {
        any = true;
        if (mant <= kMantMax) mant = mant * 10 + (uint64_t)(*p - '0');
        else { ++shift; inexact |= (*p != '0'); }
        ++p;
    }
    if (p < e && *p == '.') {
        ++p;
        while (p < e && is_digit(*p)) {
            any = true;
            if (mant <= kMantMax) { mant = mant * 10 + (uint64_t)(*p - '0'); --shift; }
            else inexact |= (*p != '0');
            ++p;
        }
    }
    if (!any) return parse_slow(tok, e, out);          // inf / nan / junk
    int e10 = 0;
    if (p < e && (*p == 'e' || *p == 'E')) {
        ++p;
        bool eneg = false;
        if (p < e && (*p == '+' || *p == '-')) { eneg = (*p == '-'); ++p; }
        if (p == e || !is_digit(*p)) return false;
        while (p < e && is_digit(*p)) {
            if (e10 < 100000) e10 = e10 * 10 + (*p - '0');
            ++p;
        }
        if (eneg) e10 = -e10;
    }
    if (p != e) return false;
    e10 += shift;
    if (!inexact && mant <= ((uint64_t)1 << 53)) {
        if (mant == 0) { *out = neg ? -0.0 : 0.0; return true; }
        double d = (double)mant;
        if (e10 >= -22 && e10 <= 22) {
            d = e10 < 0 ? d / kP10[-e10] : d * kP10[e10];
            *out = neg ? -d : d;
            return true;
        }
        if (e10 > 22 && e10 <= 22 + 15) {               // mant * 10^(e10-22) still exact below 2^53
            d *= kP10[e10 - 22];
            if (d <= 9007199254740992.0) {
                d *= kP10[22];
                *out = neg ? -d : d;
                return true;
            }
        }
    }
    return parse_slow(tok, e, out);
}

struct Range {
    size_t b0 = 0, b1 = 0;       // bytes: lines starting in [b0, b1)
    int64_t lines = 0, rows = 0; // physical lines / data lines in the range
    int64_t row0 = 0, line0 = 0; // prefix sums
    int rc = MCC_OK;
    std::string err;
};

struct Chain {
    int fd = -1;
    const char* data = nullptr;
    size_t size = 0;
    int64_t nrows = 0, ncols = 0;
    std::string path;
    std::vector<Range> ranges;
    ~Chain()
    {
        if (data && size) munmap(const_cast<char*>(data), size);
        if (fd >= 0) close(fd);
    }
};

// end of the '\n'-terminated line starting at p (index of '\n' or size): the unit thread ranges are cut at
inline size_t nl_end(const char* d, size_t p, size_t size)
{
    const void* nl = std::memchr(d + p, '\n', size - p);
    return nl ? (size_t)(static_cast<const char*>(nl) - d) : size;
}

// end of the line starting at p; a bare '\r' ends a line too (universal newlines, as np.loadtxt reads)
inline size_t line_end(const char* d, size_t p, size_t size)
{
    const size_t e = nl_end(d, p, size);
    const void* cr = std::memchr(d + p, '\r', e - p);
    return cr ? (size_t)(static_cast<const char*>(cr) - d) : e;
}

// does [p, e) hold anything but whitespace before a '#'?
inline bool has_data(const char* p, const char* e)
{
    for (; p < e; ++p) {
        if (*p == '#') return false;
        if (!is_space(*p)) return true;
    }
    return false;
}

void count_range(const Chain& c, Range& r)
{
    size_t p = r.b0;
    while (p < r.b1) {
        const size_t e = line_end(c.data, p, c.size);
        ++r.lines;
        if (has_data(c.data + p, c.data + e)) ++r.rows;
        p = e + 1;
    }
}

int count_fields(const char* p, const char* e)
{
    int n = 0;
    while (p < e) {
        while (p < e && is_space(*p)) ++p;
        if (p == e || *p == '#') break;
        ++n;
        while (p < e && !is_space(*p) && *p != '#') ++p;
    }
    return n;
}

void parse_range(const Chain& c, Range& r, double* out)
{
    size_t p = r.b0;
    int64_t row = r.row0, line = r.line0;
    const int64_t ncols = c.ncols;
    while (p < r.b1) {
        const size_t le = line_end(c.data, p, c.size);
        ++line;
        const char* q = c.data + p;
        const char* const ls = q;
        const char* e = c.data + le;
        p = le + 1;
        if (!has_data(q, e)) continue;
        double* dst = out + row * ncols;
        int64_t col = 0;
        while (q < e) {
            while (q < e && is_space(*q)) ++q;
            if (q == e || *q == '#') break;
            const char* t = q;
            while (q < e && !is_space(*q) && *q != '#') ++q;
            if (col == ncols) { ++col; break; }
            if (!parse_token(t, q, dst + col)) {
                char msg[400];
                snprintf(msg, sizeof(msg), "could not convert string '%.*s' to float64 at row %lld, column %lld (line %lld of %s)",
                         (int)std::min<ptrdiff_t>(q - t, 60), t, (long long)row, (long long)(col + 1), (long long)line, c.path.c_str());
                r.rc = MCC_ERR_PARSE;
                r.err = msg;
                return;
            }
            ++col;
        }
        if (col != ncols) {
            char msg[400];
            snprintf(msg, sizeof(msg), "the number of columns changed from %lld to %d at row %lld; line %lld of %s",
                     (long long)ncols, count_fields(ls, e), (long long)(row + 1), (long long)line, c.path.c_str());
            r.rc = MCC_ERR_RAGGED;
            r.err = msg;
            return;
        }
        ++row;
    }
}

template <class F>
void run_parallel(std::vector<Range>& ranges, F&& fn)
{
    if (ranges.size() == 1) {
        fn(ranges[0]);
        return;
    }
    std::vector<std::thread> th;
    th.reserve(ranges.size());
    for (auto& r : ranges) th.emplace_back([&fn, &r]() { fn(r); });
    for (auto& t : th) t.join();
}

}  // namespace

extern "C" {

int mce_chain_abi_version(void) { return 1; }

const char* mce_chain_last_error(void) { return g_err; }

int mce_chain_parse_token(const char* token, int64_t len, double* value)
{
    if (!token || len < 0 || !value) return fail(MCC_ERR_INVALID, "null pointer argument");
    if (!parse_token(token, token + len, value)) return fail(MCC_ERR_PARSE, "could not convert string '%.*s' to float64", (int)std::min<int64_t>(len, 60), token);
    return MCC_OK;
}

int mce_chain_open(const char* path, int32_t nthreads, void** handle, int64_t* nrows, int64_t* ncols)
{
    if (!path || !handle || !nrows || !ncols) return fail(MCC_ERR_INVALID, "null pointer argument");
    *handle = nullptr;
    Chain* c = new Chain();
    c->path = path;
    c->fd = open(path, O_RDONLY);
    if (c->fd < 0) {
        const int rc = fail(MCC_ERR_IO, "%s: %s", path, std::strerror(errno));
        delete c;
        return rc;
    }
    struct stat sb;
    if (fstat(c->fd, &sb) != 0 || !S_ISREG(sb.st_mode)) {
        const int rc = fail(MCC_ERR_IO, "%s: not a regular file", path);
        delete c;
        return rc;
    }
    c->size = (size_t)sb.st_size;
    if (c->size > 0) {
        void* m = mmap(nullptr, c->size, PROT_READ, MAP_PRIVATE, c->fd, 0);
        if (m == MAP_FAILED) {
            const int rc = fail(MCC_ERR_IO, "%s: mmap failed: %s", path, std::strerror(errno));
            c->size = 0;
            delete c;
            return rc;
        }
        c->data = static_cast<const char*>(m);
        (void)madvise(m, c->size, MADV_SEQUENTIAL);
    }
    int nt = nthreads;
    if (nt <= 0) {
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        nt = (int)std::min<size_t>(std::min<unsigned>(hw, 32u), c->size / ((size_t)4 << 20) + 1);
    }
    nt = std::max(1, std::min(nt, 256));
    // byte ranges cut at line starts
    std::vector<size_t> cuts;
    cuts.push_back(0);
    for (int t = 1; t < nt; ++t) {
        size_t b = c->size / (size_t)nt * (size_t)t;
        if (b == 0 || b >= c->size) continue;
        const size_t nl = nl_end(c->data, b - 1, c->size);        // first '\n' at or after b-1
        const size_t start = std::min(nl + 1, c->size);
        if (start > cuts.back() && start < c->size) cuts.push_back(start);
    }
    cuts.push_back(c->size);
    for (size_t i = 0; i + 1 < cuts.size(); ++i) {
        Range r;
        r.b0 = cuts[i];
        r.b1 = cuts[i + 1];
        c->ranges.push_back(r);
    }
    if (c->ranges.empty()) c->ranges.push_back(Range());
    run_parallel(c->ranges, [c](Range& r) { count_range(*c, r); });
    int64_t rows = 0, lines = 0;
    for (auto& r : c->ranges) {
        r.row0 = rows;
        r.line0 = lines;
        rows += r.rows;
        lines += r.lines;
    }
    c->nrows = rows;
    c->ncols = 0;
    if (rows > 0) {                                           // columns of the first data line
        size_t p = 0;
        while (p < c->size) {
            const size_t e = line_end(c->data, p, c->size);
            if (has_data(c->data + p, c->data + e)) {
                c->ncols = count_fields(c->data + p, c->data + e);
                break;
            }
            p = e + 1;
        }
    }
    *nrows = c->nrows;
    *ncols = c->ncols;
    *handle = c;
    return MCC_OK;
}

int mce_chain_read(void* handle, double* out)
{
    if (!handle) return fail(MCC_ERR_INVALID, "null handle");
    Chain* c = static_cast<Chain*>(handle);
    if (c->nrows == 0) return MCC_OK;
    if (!out) return fail(MCC_ERR_INVALID, "null output buffer");
    run_parallel(c->ranges, [c, out](Range& r) { parse_range(*c, r, out); });
    for (auto& r : c->ranges)                                  // first failure in file order
        if (r.rc != MCC_OK) return fail(r.rc, "%s", r.err.c_str());
    return MCC_OK;
}

void mce_chain_close(void* handle) { delete static_cast<Chain*>(handle); }

// 64-bit fingerprint of the logical dense array rows[n][d] (row stride ld doubles): the sum over its words i = r d + c of
// mix64(word_i + (salt + i) * golden) -- the function feeders.hpp's checksum_kernel computes on the device, here on the host's
// cores (order-independent integer adds, so threads take row ranges).  Multi-rank evidence() with ONE upload of the chain
// per node compares the ranks' host fingerprints instead of uploading every rank's copy to hash it on the device.
uint64_t mce_chain_fingerprint_f64(const double* rows, int64_t n, int64_t d, int64_t ld, uint64_t salt, int32_t nthreads)
{
    if (!rows || n <= 0 || d <= 0 || ld < d) return 0;
    auto mix64 = [](uint64_t z) {
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    int nt = nthreads > 0 ? nthreads : (int)std::min<int64_t>(std::max<unsigned>(1u, std::thread::hardware_concurrency()), (n * d + (1 << 18) - 1) >> 18);
    nt = std::max(1, std::min(nt, 32));
    std::vector<uint64_t> part((size_t)nt, 0);
    auto work = [&](int t) {
        const int64_t r0 = n * t / nt, r1 = n * (t + 1) / nt;
        uint64_t h = 0;
        for (int64_t r = r0; r < r1; ++r) {
            const double* row = rows + r * ld;
            for (int64_t c = 0; c < d; ++c) {
                uint64_t w;
                std::memcpy(&w, row + c, sizeof(w));
                h += mix64(w + (salt + (uint64_t)(r * d + c)) * 0x9E3779B97F4A7C15ull);
            }
        }
        part[(size_t)t] = h;
    };
    if (nt == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back(work, t);
        for (auto& x : th) x.join();
    }
    uint64_t h = 0;
    for (uint64_t v : part) h += v;
    return h;
}

}  // extern "C"
