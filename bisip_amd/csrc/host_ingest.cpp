// host_ingest.cpp -- read many 5-column spectrum files at once (the reference
// reads one file per Inversion object with np.loadtxt(skiprows=headers, delimiter=','),
// src/bisip/utils.py:121-123; format docs/user/data_format.rst:7-28).
//
// A survey of 4096 spectra is 4096 small text files: np.loadtxt costs ~0.25 ms per file, more
// than the whole sampling run of a batch on the GPU.  bisip_read_tables parses them on a few
// threads: whole file in one read, lines split by hand, numbers by std::from_chars (correctly
// rounded and locale-free, i.e. the double np.loadtxt produces; ~10 ns each, glibc's strtod took 20x that).  It understands exactly the
// documented format -- `headers` lines skipped, '#' comments, empty lines, CR LF, blanks around
// numbers, plain decimal / exponent notation, at least five columns on every row and the same
// number on each -- and flags any file that holds anything else (status 1): the caller reads
// that one with np.loadtxt, so odd files behave, and fail, exactly as in the reference.
#include <atomic>
#include <cerrno>
#include <charconv>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <unistd.h>

#include "host_precompute.h"

namespace {

inline bool is_blank(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

// One number, blanks around it allowed; false for anything but plain decimal / exponent notation
// ("nan", "inf", hex, a second sign, an empty field ...).
bool parse_field(const char *b, const char *e, double *out)
{
    while (b < e && is_blank(*b)) ++b;
    while (e > b && is_blank(e[-1])) --e;
    if (b == e) return false;
    if (*b == '+') {              // from_chars takes no leading plus; float() does
        ++b;
        if (b == e || *b == '+' || *b == '-') return false;
    }
    for (const char *p = b; p < e; ++p) {
        const char c = *p;
        if (!((c >= '0' && c <= '9') || c == '.' || c == 'e' || c == 'E' || c == '+' || c == '-')) return false;
    }
    const auto r = std::from_chars(b, e, *out, std::chars_format::general);
    return r.ec == std::errc() && r.ptr == e;
}

// 0 = table filled; 1 = not the plain format (or unreadable, or not n_rows rows): caller falls back
int read_one(const char *path, int headers, int64_t n_rows, double *out /* (n_rows, 5) */, std::vector<char> &buf)
{
    // open / read / close, no stdio stream in between (a FILE with its buffer per 2-KB file cost as much as
    // parsing it); the worker's buffer is reused from file to file
    const int fd = ::open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return 1;
    size_t have = 0;
    for (;;) {
        if (buf.size() - have < (1u << 14)) buf.resize(buf.size() * 2);
        const ssize_t got = ::read(fd, buf.data() + have, buf.size() - have);
        if (got < 0) {
            if (errno == EINTR) continue;
            ::close(fd);
            return 1;
        }
        if (got == 0) break;
        have += (size_t)got;
        if (have > (64u << 20)) { ::close(fd); return 1; }     // not a spectrum file
    }
    ::close(fd);
    const char *p = buf.data(), *end = p + have;
    int64_t line_no = 0, row = 0;
    int columns = -1;
    while (p < end) {
        const char *eol = (const char *)std::memchr(p, '\n', (size_t)(end - p));
        const char *le = eol ? eol : end;
        const char *next = eol ? eol + 1 : end;
        if (line_no++ < headers) { p = next; continue; }
        const char *hash = (const char *)std::memchr(p, '#', (size_t)(le - p));
        if (hash) le = hash;
        const char *q = le;
        if (q > p && !hash && q[-1] == '\r') --q;      // CR LF
        if (q == p) { p = next; continue; }           // empty or comment-only line: skipped, as by np.loadtxt
        q = p;
        while (q < le && is_blank(*q)) ++q;
        if (q == le) return 1;                        // blanks only: np.loadtxt sees one (bad) column there
        if (row >= n_rows) return 1;
        int col = 0;
        const char *f = p;
        for (;;) {
            const char *comma = (const char *)std::memchr(f, ',', (size_t)(le - f));
            const char *fe = comma ? comma : le;
            double v;
            if (!parse_field(f, fe, &v)) return 1;
            if (col < 5) out[row * 5 + col] = v;
            ++col;
            if (!comma) break;
            f = comma + 1;
        }
        if (col < 5 || (columns >= 0 && col != columns)) return 1;
        columns = col;
        ++row;
        p = next;
    }
    return row == n_rows ? 0 : 1;
}

}  // namespace

// bisip_read_tables (bisip_hip.hip checks the arguments)
void bisip::read_tables(const char *const *paths, int64_t n_files, int headers, int64_t n_rows,
                        double *tables, int32_t *status, int threads)
{
    if (threads < 1) threads = bisip::host_threads();
    if (threads > n_files) threads = (int)(n_files > 0 ? n_files : 1);
    std::atomic<int64_t> next{0};
    auto work = [&]() {
        std::vector<char> buf(1u << 16);
        for (;;) {
            const int64_t i = next.fetch_add(1);
            if (i >= n_files) return;
            status[i] = paths[i] ? read_one(paths[i], headers, n_rows, tables + i * n_rows * 5, buf) : 1;
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
}
