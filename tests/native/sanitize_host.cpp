// Driver for the AddressSanitizer/UBSan build of the library's host-only code
// (host_precompute.cpp, host_rng.cpp, host_ingest.cpp); built and run by tests/test_host_sanitizers.py.
// GPU sanitizers are not available on the target pool, so the host side -- the part that
// indexes caller-shaped arrays -- is checked here on the CPU.
#include <cmath>
#include <stdexcept>
#include <cstdint>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../bisip_amd/csrc/host_precompute.h"
#include "../../include/bisip_hip.h"

static int fails = 0;
#define EXPECT(c) do { if (!(c)) { std::printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

static void operands(int N, int S, int D, double c_exp)
{
    std::vector<double> w(N), taus(S), lt((size_t)D * S), zn(2 * N), err(2 * N);
    for (int j = 0; j < N; ++j) w[j] = 2 * M_PI * std::pow(10.0, 3.8 - 5.7 * (N > 1 ? (double)j / (N - 1) : 0.0));
    for (int k = 0; k < S; ++k) {
        const double l = -6.0 + 8.0 * (S > 1 ? (double)k / (S - 1) : 0.0);
        taus[k] = std::pow(10.0, l);
        for (int p = 0; p < D; ++p) lt[(size_t)p * S + k] = std::pow(l, p);
    }
    for (int i = 0; i < 2 * N; ++i) { zn[i] = (i < N ? 0.8 : -0.05) + 0.01 * std::sin(i); err[i] = 0.002 + 1e-4 * (i % 7); }
    bisip::PolyDecompOperands o;
    bisip::polydecomp_operands(N, w.data(), S, taus.data(), D, lt.data(), c_exp, zn.data(), err.data(), o);
    const int n = D + 1;
    EXPECT(o.N == N && o.S == S && o.D == D);
    EXPECT((int)o.G_re.size() == N * D && (int)o.G_im.size() == N * D);
    EXPECT((int)o.R.size() == n * n && (int)o.bhat.size() == n && (int)o.e.size() == n);
    EXPECT(std::isfinite(o.rest) && o.rest >= -1e-9);
    for (double v : o.R) EXPECT(std::isfinite(v));
    for (int i = 1; i < n; ++i) for (int j = 0; j < i; ++j) EXPECT(o.R[(size_t)i * n + j] == 0.0);
    std::vector<double> lnw, iv;
    bisip::common_operands(N, w.data(), err.data(), lnw, iv);
    EXPECT((int)lnw.size() == N && (int)iv.size() == 2 * N);
    EXPECT(std::isfinite(bisip::loglike_const(2 * N, err.data())));
    // geometric-grid detection (kernels.h: BOUNDS_GRID): whatever w is, dlnw is set; a log-spaced list of
    // >= 8 frequencies qualifies, the same list with one entry moved or fewer entries does not
    double dlnw = -1.0;
    const bool as_given = bisip::grid_step(N, w.data(), lnw.data(), &dlnw);
    EXPECT(std::isfinite(dlnw) && (as_given || dlnw == 0.0));
    std::vector<double> geo((size_t)(N > 0 ? N : 1)), lg;
    for (int j = 0; j < N; ++j) geo[(size_t)j] = 0.07 * std::pow(1.7, j);
    bisip::common_operands(N, geo.data(), err.data(), lg, iv);
    EXPECT(bisip::grid_step(N, geo.data(), lg.data(), &dlnw) == (N >= 8));
    if (N >= 8) {
        EXPECT(std::fabs(dlnw - std::log(1.7)) < 1e-14);
        geo[(size_t)N / 2] *= 1.0001;
        bisip::common_operands(N, geo.data(), err.data(), lg, iv);
        EXPECT(!bisip::grid_step(N, geo.data(), lg.data(), &dlnw) && dlnw == 0.0);
    }
}

static void stream(int64_t W, int64_t n_steps)
{
    std::vector<uint32_t> key(624);
    uint32_t s = 5489u + (uint32_t)W;
    for (int i = 0; i < 624; ++i) { key[i] = s; s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1; }
    int32_t pos = 624;
    const int64_t nh = (W + 1) / 2, tot = n_steps * 2 * nh;
    std::vector<int32_t> act(tot), par(tot);
    std::vector<double> zz(tot), u(tot);
    EXPECT(bisip_numpy_stretch_stream(key.data(), &pos, W, 2.0, n_steps, act.data(), par.data(), zz.data(), u.data()) == 0);
    EXPECT(pos >= 0 && pos <= 624);
    for (int64_t k = 0; k < n_steps; ++k) {
        std::vector<int> seen(W, 0);
        for (int h = 0; h < 2; ++h) {
            const int64_t Ns = h ? W / 2 : nh;
            for (int64_t t = 0; t < Ns; ++t) {
                const int64_t i = (k * 2 + h) * nh + t;
                EXPECT(act[i] >= 0 && act[i] < W && par[i] >= 0 && par[i] < W);
                EXPECT(zz[i] >= 0.5 && zz[i] <= 2.0 && u[i] >= 0.0 && u[i] < 1.0);
                seen[act[i]] += 1;
            }
        }
        for (int64_t i = 0; i < W; ++i) EXPECT(seen[i] == 1);   // every walker is active exactly once per iteration
    }
}

// the file parser on well-formed and hostile inputs (it must flag, never read out of bounds)
static void ingest(const char *dir)
{
    struct Case { const char *name; std::string text; int want; };
    std::string good = "freq,amp,pha,amp_err,pha_err\n";
    for (int r = 0; r < 4; ++r) good += "1.5e3, 2.25 ,-3,+4.0,5e-1\r\n";
    const Case cases[] = {
        {"good.csv", good, 0},
        {"no_newline.csv", "h\n1,2,3,4,5\n1,2,3,4,5\n1,2,3,4,5\n1,2,3,4,5", 0},
        {"comments.csv", "h\n# c\n1,2,3,4,5 # x\n\n1,2,3,4,5\n1,2,3,4,5\n1,2,3,4,5\n", 0},
        {"extra_cols.csv", "h\n1,2,3,4,5,6,7\n1,2,3,4,5,6,7\n1,2,3,4,5,6,7\n1,2,3,4,5,6,7\n", 0},
        {"empty.csv", "", 1},
        {"only_header.csv", "h\n", 1},
        {"short_row.csv", "h\n1,2,3,4\n1,2,3,4,5\n1,2,3,4,5\n1,2,3,4,5\n", 1},
        {"ragged.csv", "h\n1,2,3,4,5\n1,2,3,4,5,6\n1,2,3,4,5\n1,2,3,4,5\n", 1},
        {"too_many_rows.csv", "h\n1,2,3,4,5\n1,2,3,4,5\n1,2,3,4,5\n1,2,3,4,5\n1,2,3,4,5\n", 1},
        {"words.csv", "h\n1,2,3,4,nan\n1,2,3,4,5\n1,2,3,4,5\n1,2,3,4,5\n", 1},
        {"hex.csv", "h\n0x10,2,3,4,5\n1,2,3,4,5\n1,2,3,4,5\n1,2,3,4,5\n", 1},
        {"empty_field.csv", "h\n1,,3,4,5\n1,2,3,4,5\n1,2,3,4,5\n1,2,3,4,5\n", 1},
        {"trailing_comma.csv", "h\n1,2,3,4,5,\n1,2,3,4,5,\n1,2,3,4,5,\n1,2,3,4,5,\n", 1},
        {"blanks.csv", "h\n1,2,3,4,5\n   \n1,2,3,4,5\n1,2,3,4,5\n1,2,3,4,5\n", 1},
        {"signs.csv", "h\n+-1,2,3,4,5\n1,2,3,4,5\n1,2,3,4,5\n1,2,3,4,5\n", 1},
        {"huge_number.csv", "h\n1e999999,2,3,4,5\n1,2,3,4,5\n1,2,3,4,5\n1,2,3,4,5\n", 1},
        {"binary.csv", std::string("h\n") + std::string(3, '\0') + ",\xff\xfe,3,4,5\n1,2,3,4,5\n1,2,3,4,5\n1,2,3,4,5\n", 1},
    };
    std::vector<std::string> paths;
    std::vector<int> want;
    for (const Case &c : cases) {
        const std::string p = std::string(dir) + "/" + c.name;
        FILE *f = std::fopen(p.c_str(), "wb");
        EXPECT(f != nullptr);
        if (!f) continue;
        std::fwrite(c.text.data(), 1, c.text.size(), f);
        std::fclose(f);
        paths.push_back(p);
        want.push_back(c.want);
    }
    paths.push_back(std::string(dir) + "/does_not_exist.csv");
    want.push_back(1);
    std::vector<const char *> cp;
    for (auto &p : paths) cp.push_back(p.c_str());
    for (int threads : {1, 3, 0}) {
        std::vector<double> tables(cp.size() * 4 * 5, -1.0);
        std::vector<int32_t> status(cp.size(), 7);
        bisip::read_tables(cp.data(), (int64_t)cp.size(), 1, 4, tables.data(), status.data(), threads);
        for (size_t i = 0; i < cp.size(); ++i) EXPECT(status[i] == want[i]);
        EXPECT(tables[0] == 1500.0 && tables[1] == 2.25 && tables[2] == -3.0 && tables[3] == 4.0 && tables[4] == 0.5);
        EXPECT(tables[3 * 4 * 5 + 4] == 5.0);          // extra columns ignored
    }
}

// blocks of work on host threads: every index exactly once; an exception comes back to the caller
static void blocks()
{
    for (int64_t n : {0, 1, 7, 64, 1000}) {
        std::vector<std::atomic<int>> hit((size_t)(n > 0 ? n : 1));
        for (auto &h : hit) h = 0;
        bisip::parallel_blocks(n, 3, [&](int64_t lo, int64_t hi) { for (int64_t i = lo; i < hi; ++i) hit[(size_t)i]++; });
        for (int64_t i = 0; i < n; ++i) EXPECT(hit[(size_t)i] == 1);
    }
    bool thrown = false;
    try {
        bisip::parallel_blocks(1000, 1, [&](int64_t lo, int64_t) { if (lo > 0) throw std::runtime_error("x"); });
    } catch (const std::runtime_error &) { thrown = true; }
    EXPECT(thrown || bisip::host_threads() == 1);
}

// the probing that decides which reduced kernel runs, and the after-the-fact reference
static void reduced(int N, int S, int D)
{
    std::vector<double> w(N), taus(S), lt((size_t)D * S), zn(2 * N), err(2 * N);
    for (int j = 0; j < N; ++j) w[j] = 2 * M_PI * std::pow(10.0, 3.8 - 5.7 * (N > 1 ? (double)j / (N - 1) : 0.0));
    for (int k = 0; k < S; ++k) {
        const double l = -6.0 + 8.0 * (S > 1 ? (double)k / (S - 1) : 0.0);
        taus[k] = std::pow(10.0, l);
        for (int p = 0; p < D; ++p) lt[(size_t)p * S + k] = std::pow(l, p);
    }
    for (int i = 0; i < 2 * N; ++i) { zn[i] = (i < N ? 0.8 : -0.05) + 0.01 * std::sin(i); err[i] = 0.002 + 1e-4 * (i % 7); }
    bisip::PolyDecompOperands o;
    bisip::polydecomp_kernel_sums(N, w.data(), S, taus.data(), D, lt.data(), 1.0, o);
    bisip::polydecomp_reduce(zn.data(), err.data(), o);
    const int n = D + 1;
    std::vector<double> lo(n, -1.0), hi(n, 1.0), bh(n), e(n), el(n);
    lo[0] = 0.9; hi[0] = 1.1;
    const double lconst = bisip::loglike_const(2 * N, err.data());
    for (bool comp : {false, true}) {
        const double est = bisip::reduced_center(n, o.R, o.Rl, o.qty, o.bhat_ls, o.rest, lconst, lo.data(), hi.data(), comp, bh.data(), e.data(), el.data());
        EXPECT(est >= 0.0 || est != est);
    }
    std::vector<double> th(n, 0.01);
    th[0] = 1.0;
    EXPECT(std::isfinite(bisip::reduced_logp_reference(n, o.Rl, o.qty, o.rest, lconst, th.data())));
    // the same spectrum as a context keeps it: shared probes, the plain tier, then the compensated tier on
    // operands from the QR in binary128, and the yardstick from those operands
    bisip::ReducedProblem p;
    bisip::reduced_from_operands(o, lconst, p);
    bisip::ReducedProbes probes;
    bisip::reduced_probes(p, lo.data(), hi.data(), probes);
    EXPECT(probes.n == n && probes.rows.size() == probes.count() * (size_t)n && probes.n_regular <= probes.count());
    const double plain = bisip::reduced_center_plain(p, probes, lo.data(), hi.data(), 0.05, bh.data(), e.data(), el.data());
    EXPECT(plain >= 0.0 || plain != plain);
    const double ld_ref = bisip::reduced_logp_reference(p, th.data());
    if (2 * N >= n) {
        bisip::reduced_make_quad(*bisip::polydecomp_kernel_sums_quad(N, w.data(), S, taus.data(), D, lt.data(), 1.0), zn.data(), err.data(), p);
        EXPECT(p.has_quad() && (int)p.Rc.size() == n * n && (int)p.Rc_lo.size() == n * n);
        const double comp = bisip::reduced_center_comp(p, probes, lo.data(), hi.data(), 0.05, bh.data(), e.data(), el.data());
        EXPECT(comp >= 0.0 || comp != comp);
        const double q_ref = bisip::reduced_logp_reference(p, th.data());
        EXPECT(std::isfinite(q_ref) && std::fabs(q_ref - ld_ref) <= 1e-6 * (1.0 + std::fabs(ld_ref)));
    }
}

int main(int argc, char **argv)
{
    if (argc > 1) ingest(argv[1]);
    blocks();
    for (int N : {2, 20}) for (int D : {1, 6, 11}) reduced(N, 2 * N, D);
    for (int N : {1, 2, 20, 33}) for (int S : {1, 7, 40}) for (int D : {1, 6, 11})
        for (double c : {1.0, 0.5}) operands(N, S, D, c);
    for (int64_t W : {2, 3, 32, 33, 1000}) stream(W, 7);
    stream(8, 0);
    int32_t pos = 0; uint32_t key[624] = {0};
    EXPECT(bisip_numpy_stretch_stream(key, &pos, 1, 2.0, 1, nullptr, nullptr, nullptr, nullptr) != 0);   // bad arguments
    std::printf(fails ? "sanitize_host: %d failures\n" : "sanitize_host: ok\n", fails);
    return fails ? 1 : 0;
}
