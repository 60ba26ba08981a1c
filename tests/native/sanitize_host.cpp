// Driver for the AddressSanitizer/UBSan build of the library's host-only code
// (host_precompute.cpp, host_rng.cpp); built and run by tests/test_host_sanitizers.py.
// GPU sanitizers are not available on the target pool, so the host side -- the part that
// indexes caller-shaped arrays -- is checked here on the CPU.
#include <cmath>
#include <cstdint>
#include <cstdio>
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

int main()
{
    for (int N : {1, 2, 20, 33}) for (int S : {1, 7, 40}) for (int D : {1, 6, 11})
        for (double c : {1.0, 0.5}) operands(N, S, D, c);
    for (int64_t W : {2, 3, 32, 33, 1000}) stream(W, 7);
    stream(8, 0);
    int32_t pos = 0; uint32_t key[624] = {0};
    EXPECT(bisip_numpy_stretch_stream(key, &pos, 1, 2.0, 1, nullptr, nullptr, nullptr, nullptr) != 0);   // bad arguments
    std::printf(fails ? "sanitize_host: %d failures\n" : "sanitize_host: ok\n", fails);
    return fails ? 1 : 0;
}
