// Host self-test of the in-register FFT building blocks against a direct O(N^2) DFT in double.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "../../speech_to_text_amd/csrc/fft_core.h"

template <int R>
static double check_rdft_cols() {
    const double PI = 3.14159265358979323846;
    double worst = 0.0;
    for (int trial = 0; trial < 50; ++trial) {
        float r[2 * R], ar[R], ai[R]; double rd[2 * R];
        for (int n = 0; n < 2 * R; ++n) rd[n] = r[n] = (float)rand() / RAND_MAX - 0.5f;
        fc::rdft_cols<R>(r, ar, ai);
        double ev = 0, od = 0;
        for (int n = 0; n < 2 * R; ++n) ((n & 1) ? od : ev) += rd[n];
        worst = fmax(worst, fabs(ev - ar[0]));
        worst = fmax(worst, fabs(od - ai[0]));
        for (int k = 1; k < R; ++k) {
            double sr = 0, si = 0;
            for (int n = 0; n < 2 * R; ++n) {
                sr += rd[n] * cos(2 * PI * n * k / (2 * R)); si -= rd[n] * sin(2 * PI * n * k / (2 * R));
            }
            const double sc = (k == R / 2) ? 1.0 : 2.0;
            worst = fmax(worst, fabs(sc * sr - ar[k]));
            worst = fmax(worst, fabs(sc * si - ai[k]));
        }
    }
    return worst;
}

static double check_rdft25() {
    const double PI = 3.14159265358979323846;
    double worst = 0.0;
    for (int trial = 0; trial < 50; ++trial) {
        float x[25], yr[13], yi[13]; double xd[25];
        for (int n = 0; n < 25; ++n) xd[n] = x[n] = (float)rand() / RAND_MAX - 0.5f;
        fc::rdft25(x, yr, yi);
        for (int k = 0; k < 13; ++k) {
            double sr = 0, si = 0;
            for (int n = 0; n < 25; ++n) { sr += xd[n] * cos(2 * PI * n * k / 25); si -= xd[n] * sin(2 * PI * n * k / 25); }
            worst = fmax(worst, fabs(sr - yr[k]));
            worst = fmax(worst, fabs(si - yi[k]));
        }
    }
    return worst;
}

template <int A>
static double check_rdft_odd() {
    const double PI = 3.14159265358979323846;
    double worst = 0.0;
    for (int trial = 0; trial < 50; ++trial) {
        float x[A], yr[A / 2 + 1], yi[A / 2 + 1]; double xd[A];
        for (int n = 0; n < A; ++n) xd[n] = x[n] = (float)rand() / RAND_MAX - 0.5f;
        fc::rdft_odd<A>(x, yr, yi);
        for (int k = 0; k <= A / 2; ++k) {
            double sr = 0, si = 0;
            for (int n = 0; n < A; ++n) { sr += xd[n] * cos(2 * PI * n * k / A); si -= xd[n] * sin(2 * PI * n * k / A); }
            worst = fmax(worst, fabs(sr - yr[k]));
            worst = fmax(worst, fabs(si - yi[k]));
        }
    }
    return worst;
}

template <int A>
static double check_rdft_even() {
    const double PI = 3.14159265358979323846;
    double worst = 0.0;
    for (int trial = 0; trial < 50; ++trial) {
        float x[A], yr[A / 2 + 1], yi[A / 2 + 1]; double xd[A];
        for (int n = 0; n < A; ++n) xd[n] = x[n] = (float)rand() / RAND_MAX - 0.5f;
        fc::rdft_even<A>(x, yr, yi);
        for (int k = 0; k <= A / 2; ++k) {
            double sr = 0, si = 0;
            for (int n = 0; n < A; ++n) { sr += xd[n] * cos(2 * PI * n * k / A); si -= xd[n] * sin(2 * PI * n * k / A); }
            worst = fmax(worst, fabs(sr - yr[k]));
            worst = fmax(worst, fabs(si - yi[k]));
        }
    }
    return worst;
}

int main() {
    const double PI = 3.14159265358979323846;
    double worst = 0.0;
    srand(7);
    worst = fmax(worst, check_rdft25());
    // every even length the 16 A front end is built for
    worst = fmax(worst, check_rdft_even<8>()); worst = fmax(worst, check_rdft_even<10>());
    worst = fmax(worst, check_rdft_even<12>()); worst = fmax(worst, check_rdft_even<14>());
    worst = fmax(worst, check_rdft_even<18>()); worst = fmax(worst, check_rdft_even<20>());
    worst = fmax(worst, check_rdft_even<22>()); worst = fmax(worst, check_rdft_even<24>());
    worst = fmax(worst, check_rdft_even<26>()); worst = fmax(worst, check_rdft_even<28>());
    worst = fmax(worst, check_rdft_even<30>()); worst = fmax(worst, check_rdft_even<34>());
    worst = fmax(worst, check_rdft_even<36>()); worst = fmax(worst, check_rdft_even<38>());
    worst = fmax(worst, check_rdft_even<40>()); worst = fmax(worst, check_rdft_even<42>());
    worst = fmax(worst, check_rdft_even<44>()); worst = fmax(worst, check_rdft_even<46>());
    worst = fmax(worst, check_rdft_even<48>()); worst = fmax(worst, check_rdft_even<50>());
    worst = fmax(worst, check_rdft_even<52>()); worst = fmax(worst, check_rdft_even<54>());
    worst = fmax(worst, check_rdft_even<56>()); worst = fmax(worst, check_rdft_even<58>());
    worst = fmax(worst, check_rdft_even<60>()); worst = fmax(worst, check_rdft_even<62>());
    // every odd length the 16 A front end is built for (25 has a transform of its own, checked above)
    worst = fmax(worst, check_rdft_odd<9>()); worst = fmax(worst, check_rdft_odd<11>());
    worst = fmax(worst, check_rdft_odd<13>()); worst = fmax(worst, check_rdft_odd<15>());
    worst = fmax(worst, check_rdft_odd<17>()); worst = fmax(worst, check_rdft_odd<19>());
    worst = fmax(worst, check_rdft_odd<21>()); worst = fmax(worst, check_rdft_odd<23>());
    worst = fmax(worst, check_rdft_odd<25>()); worst = fmax(worst, check_rdft_odd<27>());
    worst = fmax(worst, check_rdft_odd<29>()); worst = fmax(worst, check_rdft_odd<31>());
    worst = fmax(worst, check_rdft_odd<33>()); worst = fmax(worst, check_rdft_odd<35>());
    worst = fmax(worst, check_rdft_odd<37>()); worst = fmax(worst, check_rdft_odd<39>());
    worst = fmax(worst, check_rdft_odd<41>()); worst = fmax(worst, check_rdft_odd<43>());
    worst = fmax(worst, check_rdft_odd<45>()); worst = fmax(worst, check_rdft_odd<47>());
    worst = fmax(worst, check_rdft_odd<49>()); worst = fmax(worst, check_rdft_odd<51>());
    worst = fmax(worst, check_rdft_odd<53>()); worst = fmax(worst, check_rdft_odd<55>());
    worst = fmax(worst, check_rdft_odd<57>()); worst = fmax(worst, check_rdft_odd<59>());
    worst = fmax(worst, check_rdft_odd<61>()); worst = fmax(worst, check_rdft_odd<63>());
    // the series behind the trigonometric constants of lengths without a literal table: within
    // 5e-15 of libm everywhere, and float for float what the literal tables hold
    for (int A = 2; A <= 64; ++A)
        for (int j = 0; j < A; ++j) {
            worst = fmax(worst, fabs(fc::trig_unit(j, A, false) - cos(2 * PI * j / A)) > 5e-15 ? 1.0 : 0.0);
            worst = fmax(worst, fabs(fc::trig_unit(j, A, true) - sin(2 * PI * j / A)) > 5e-15 ? 1.0 : 0.0);
        }
    for (int j = 0; j < 20; ++j)
        if ((float)fc::trig_unit(j, 20, false) != fc::C20[j] || (float)fc::trig_unit(j, 20, true) != fc::S20[j]) worst = 1.0;
    for (int j = 0; j < 60; ++j)
        if ((float)fc::trig_unit(j, 60, false) != fc::C60[j] || (float)fc::trig_unit(j, 60, true) != fc::S60[j]) worst = 1.0;
    worst = fmax(worst, check_rdft_cols<2>());
    worst = fmax(worst, check_rdft_cols<4>());
    worst = fmax(worst, check_rdft_cols<8>());
    for (int trial = 0; trial < 50; ++trial) {
        float re[16], im[16]; double xr[16], xi[16];
        for (int n = 0; n < 16; ++n) {
            xr[n] = re[n] = (float)rand() / RAND_MAX - 0.5f;
            xi[n] = im[n] = (float)rand() / RAND_MAX - 0.5f;
        }
        {   // the windowed form equals cfft16 of the windowed, pair-packed sequence
            float x[32], w[32], wr[16], wi[16], zr[16], zi[16];
            for (int n = 0; n < 32; ++n) { x[n] = (float)rand() / RAND_MAX - 0.5f; w[n] = (float)rand() / RAND_MAX; }
            for (int m = 0; m < 16; ++m) { zr[m] = x[2 * m] * w[2 * m]; zi[m] = x[2 * m + 1] * w[2 * m + 1]; }
            fc::cfft16(zr, zi);
            fc::cfft16_windowed(x, w, wr, wi);
            for (int k = 0; k < 16; ++k) { worst = fmax(worst, fabs(zr[k] - wr[k])); worst = fmax(worst, fabs(zi[k] - wi[k])); }
        }
        fc::cfft16(re, im);
        for (int k = 0; k < 16; ++k) {
            double sr = 0, si = 0;
            for (int n = 0; n < 16; ++n) {
                const double c = cos(2 * PI * n * k / 16), s = -sin(2 * PI * n * k / 16);
                sr += xr[n] * c - xi[n] * s; si += xr[n] * s + xi[n] * c;
            }
            worst = fmax(worst, fabs(sr - re[fc::cfft16_pos(k)]));
            worst = fmax(worst, fabs(si - im[fc::cfft16_pos(k)]));
        }
        float r[32], ar[16], ai[16]; double rd[32];
        for (int n = 0; n < 32; ++n) rd[n] = r[n] = (float)rand() / RAND_MAX - 0.5f;
        fc::rdft32(r, ar, ai);
        double A0 = 0, A16 = 0;
        for (int n = 0; n < 32; ++n) { A0 += rd[n]; A16 += (n & 1) ? -rd[n] : rd[n]; }
        worst = fmax(worst, fabs((A0 + A16) / 2 - ar[0]));
        worst = fmax(worst, fabs((A0 - A16) / 2 - ai[0]));
        for (int k = 1; k < 16; ++k) {
            double sr = 0, si = 0;
            for (int n = 0; n < 32; ++n) {
                sr += rd[n] * cos(2 * PI * n * k / 32); si -= rd[n] * sin(2 * PI * n * k / 32);
            }
            const double sc = (k == 8) ? 1.0 : 2.0;
            worst = fmax(worst, fabs(sc * sr - ar[k]));
            worst = fmax(worst, fabs(sc * si - ai[k]));
        }
    }
    printf("max_abs_err %.3e\n", worst);
    return worst < 5e-6 ? 0 : 1;
}
