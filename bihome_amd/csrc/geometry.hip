// Per-sample dense algebra of the biHomE head: 4-point homography solve, DLT on sampled
// correspondences (one wavefront per problem: shuffle reductions for the Hartley statistics and the
// A^T A blocks, a 9x9 Jacobi eigen-solve in LDS), their adjoints, and DSAC reprojection scoring.
// All arithmetic in double: these problems are tiny (B*n <= a few hundred) and latency-bound.
#include "common.h"

// ---------------------------------------------------------------------------------------------
// 8x8 solve with partial pivoting; S is an 8x9 augmented system in LDS, row stride 9
// ---------------------------------------------------------------------------------------------
__device__ static void solve8(double* S, double* x) {
    for (int k = 0; k < 8; ++k) {
        int r = k;
        double best = fabs(S[k * 9 + k]);
        for (int i = k + 1; i < 8; ++i) {
            double v = fabs(S[i * 9 + k]);
            if (v > best) { best = v; r = i; }
        }
        if (r != k)
            for (int j = k; j < 9; ++j) { double t = S[k * 9 + j]; S[k * 9 + j] = S[r * 9 + j]; S[r * 9 + j] = t; }
        double inv = 1.0 / S[k * 9 + k];
        for (int i = k + 1; i < 8; ++i) {
            double f = S[i * 9 + k] * inv;
            for (int j = k + 1; j < 9; ++j) S[i * 9 + j] -= f * S[k * 9 + j];
        }
    }
    for (int k = 7; k >= 0; --k) {
        double acc = S[k * 9 + 8];
        for (int j = k + 1; j < 8; ++j) acc -= S[k * 9 + j] * x[j];
        x[k] = acc / S[k * 9 + k];
    }
}

__device__ static void corner_xy(int i, double W, double H, double& x, double& y) {
    // image_shape_to_corners: [[0,0],[W,0],[W,H],[0,H]]
    x = (i == 1 || i == 2) ? W : 0.0;
    y = (i >= 2) ? H : 0.0;
}

__global__ void __launch_bounds__(64) h4pt_fwd_kernel(const float* __restrict__ delta, int B, float W, float H,
                                                      double* __restrict__ H64, float* __restrict__ H32) {
    __shared__ double sm[64 * 73];
    int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    double* S = sm + threadIdx.x * 73;
    for (int i = 0; i < 4; ++i) {
        double x, y;
        corner_xy(i, W, H, x, y);
        double u = x + (double)delta[b * 8 + 2 * i], v = y + (double)delta[b * 8 + 2 * i + 1];
        double* r0 = S + (2 * i) * 9;
        double* r1 = S + (2 * i + 1) * 9;
        r0[0] = x; r0[1] = y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -x * u; r0[7] = -y * u; r0[8] = u;
        r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = x; r1[4] = y; r1[5] = 1; r1[6] = -x * v; r1[7] = -y * v; r1[8] = v;
    }
    double hsol[8];
    solve8(S, hsol);
    for (int j = 0; j < 8; ++j) {
        H64[b * 9 + j] = hsol[j];
        if (H32) H32[b * 9 + j] = (float)hsol[j];
    }
    H64[b * 9 + 8] = 1.0;
    if (H32) H32[b * 9 + 8] = 1.0f;
}

__global__ void __launch_bounds__(64) h4pt_bwd_kernel(const float* __restrict__ delta, const double* __restrict__ H64,
                                                      const double* __restrict__ gH, int B, float W, float H,
                                                      float* __restrict__ gdelta) {
    __shared__ double sm[64 * 73];
    int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    double* S = sm + threadIdx.x * 73;
    // build A^T | g : S[row j][col i] = A[i][j]
    for (int i = 0; i < 4; ++i) {
        double x, y;
        corner_xy(i, W, H, x, y);
        double u = x + (double)delta[b * 8 + 2 * i], v = y + (double)delta[b * 8 + 2 * i + 1];
        double r0[8] = {x, y, 1, 0, 0, 0, -x * u, -y * u};
        double r1[8] = {0, 0, 0, x, y, 1, -x * v, -y * v};
        for (int j = 0; j < 8; ++j) { S[j * 9 + 2 * i] = r0[j]; S[j * 9 + 2 * i + 1] = r1[j]; }
    }
    for (int j = 0; j < 8; ++j) S[j * 9 + 8] = gH[b * 9 + j];
    double lam[8];
    solve8(S, lam);
    double h6 = H64[b * 9 + 6], h7 = H64[b * 9 + 7];
    for (int i = 0; i < 4; ++i) {
        double x, y;
        corner_xy(i, W, H, x, y);
        double f = 1.0 + x * h6 + y * h7;     // dL/du_i = lam_{2i} (1 + x h6 + y h7)
        gdelta[b * 8 + 2 * i] = (float)(lam[2 * i] * f);
        gdelta[b * 8 + 2 * i + 1] = (float)(lam[2 * i + 1] * f);
    }
}

// ---------------------------------------------------------------------------------------------
// DLT: one wave per (sample, hypothesis)
// ---------------------------------------------------------------------------------------------
struct Hartley {
    double mx, my, s, dbar;
};

// statistics of the P points held by this wave (each lane passes its partial sums)
__device__ static Hartley hartley_stats(const double* px, const double* py, int cnt, int P) {
    double sx = 0, sy = 0;
    for (int k = 0; k < cnt; ++k) { sx += px[k]; sy += py[k]; }
    sx = wave_sum(sx); sy = wave_sum(sy);
    Hartley t;
    t.mx = sx / P; t.my = sy / P;
    double sd = 0;
    for (int k = 0; k < cnt; ++k) { double dx = px[k] - t.mx, dy = py[k] - t.my; sd += sqrt(dx * dx + dy * dy); }
    sd = wave_sum(sd);
    t.dbar = sd / P;
    t.s = 1.4142135623730951 / (t.dbar + 1e-8);
    return t;
}

#define DLT_MAXPTS 8   // points per lane: P <= 512

// Jacobi eigen-decomposition of the symmetric 9x9 matrix in LDS A (destroyed); V gets eigenvectors in columns.
// Parallel (round-robin) ordering: round r of a sweep rotates the four disjoint pairs {(r+k) mod 9, (r-k) mod 9},
// k = 1..4 (index r sits out; every pair {a, b} occurs once per sweep, in the round with 2r = a+b mod 9), so a sweep is
// 9 dependent steps instead of 36.  The rotations of a round commute (disjoint index pairs): first A J and V J
// (lane = (row, pair), columns p and q), then J^T (A J) (lane = (column, pair), rows p and q).
__device__ static void jacobi9(double* A, double* V, int lane) {
    if (lane < 9)
        for (int j = 0; j < 9; ++j) V[lane * 9 + j] = (lane == j) ? 1.0 : 0.0;
    __syncthreads();
    const int k = lane >> 2, pr = lane & 3;       // lanes 0..35: row / column k, pair pr
    for (int sweep = 0; sweep < 16; ++sweep) {
        // convergence: off-diagonal mass vs diagonal mass
        double off = 0, dia = 0;
        if (lane < 9)
            for (int j = 0; j < 9; ++j) { double v = A[lane * 9 + j]; if (j == lane) dia += v * v; else off += v * v; }
        off = wave_sum(off); dia = wave_sum(dia);
        if (off <= 1e-30 * dia || off == 0.0) break;      // off-diagonal Frobenius mass below 1e-15 of the diagonal: converged in double
        for (int r = 0; r < 9; ++r) {
            int ia = r + pr + 1, ib = r + 8 - pr;
            ia = ia >= 9 ? ia - 9 : ia; ib = ib >= 9 ? ib - 9 : ib;
            const int p = min(ia, ib), q = max(ia, ib);
            const double apq = A[p * 9 + q], app = A[p * 9 + p], aqq = A[q * 9 + q];
            double c = 1.0, s = 0.0;
            if (fabs(apq) > 1e-300) {
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                c = 1.0 / sqrt(t * t + 1.0); s = t * c;
            }
            __syncthreads();                      // every lane has read its pair's entries
            if (k < 9) {
                const double akp = A[k * 9 + p], akq = A[k * 9 + q];
                A[k * 9 + p] = c * akp - s * akq;
                A[k * 9 + q] = s * akp + c * akq;
                const double vkp = V[k * 9 + p], vkq = V[k * 9 + q];
                V[k * 9 + p] = c * vkp - s * vkq;
                V[k * 9 + q] = s * vkp + c * vkq;
            }
            __syncthreads();
            if (k < 9) {
                const double apk = A[p * 9 + k], aqk = A[q * 9 + k];
                A[p * 9 + k] = c * apk - s * aqk;
                A[q * 9 + k] = s * apk + c * aqk;
            }
            __syncthreads();
        }
    }
    __syncthreads();
}

// loads this wave's points; returns count for this lane
__device__ static int load_points(const float* pf, const int64_t* choice, int b, int j, int n, int P, int h, int w, int lane,
                                  double* x1, double* y1, double* x2, double* y2, int* idx) {
    int cnt = 0;
    const float* pfx = pf + (size_t)b * 2 * h * w;
    const float* pfy = pfx + (size_t)h * w;
    for (int p = lane; p < P && cnt < DLT_MAXPTS; p += 64, ++cnt) {
        int id = (int)choice[(size_t)b * ((size_t)P * n) + (size_t)j * P + p];
        idx[cnt] = id;
        double cx = (double)(id % w), cy = (double)(id / w);
        x1[cnt] = cx; y1[cnt] = cy;
        x2[cnt] = cx + (double)pfx[id];
        y2[cnt] = cy + (double)pfy[id];
    }
    return cnt;
}

// grid: (B, n); block 64
__global__ void __launch_bounds__(64) dlt_fwd_kernel(const float* __restrict__ pf, const int64_t* __restrict__ choice,
                                                     int P, int h, int w, float* __restrict__ Hout,
                                                     float* __restrict__ delta_hat, double* __restrict__ eig) {
    __shared__ double A[81];
    __shared__ double V[81];
    const int b = blockIdx.x, j = blockIdx.y, n = gridDim.y, lane = threadIdx.x;
    const int prob = b * n + j;
    double x1[DLT_MAXPTS], y1[DLT_MAXPTS], x2[DLT_MAXPTS], y2[DLT_MAXPTS];
    int idx[DLT_MAXPTS];
    int cnt = load_points(pf, choice, b, j, n, P, h, w, lane, x1, y1, x2, y2, idx);
    Hartley t1 = hartley_stats(x1, y1, cnt, P);
    Hartley t2 = hartley_stats(x2, y2, cnt, P);
    // block sums: S0 = a a^T, Sx = x2 a a^T, Sy = y2 a a^T, Sr = (x2^2+y2^2) a a^T with a = [x1 y1 1] (normalised)
    double acc[24];
    for (int i = 0; i < 24; ++i) acc[i] = 0;
    for (int k = 0; k < cnt; ++k) {
        double a0 = t1.s * (x1[k] - t1.mx), a1 = t1.s * (y1[k] - t1.my), a2 = 1.0;
        double u = t2.s * (x2[k] - t2.mx), v = t2.s * (y2[k] - t2.my);
        double aa[6] = {a0 * a0, a0 * a1, a0 * a2, a1 * a1, a1 * a2, a2 * a2};
        double r = u * u + v * v;
        for (int i = 0; i < 6; ++i) {
            acc[i] += aa[i]; acc[6 + i] += u * aa[i]; acc[12 + i] += v * aa[i]; acc[18 + i] += r * aa[i];
        }
    }
    for (int i = 0; i < 24; ++i) acc[i] = wave_sum(acc[i]);
    if (lane == 0) {
        // symmetric 3x3 from 6 uniques: index map
        const int sym[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                int s = sym[r][c];
                A[(r) * 9 + c] = acc[s];                 // M00  (ay: [a,0,-x2 a])
                A[(3 + r) * 9 + 3 + c] = acc[s];         // M11  (ax: [0,-a,y2 a])
                A[(r) * 9 + 3 + c] = 0; A[(3 + r) * 9 + c] = 0;
                A[(r) * 9 + 6 + c] = -acc[6 + s]; A[(6 + r) * 9 + c] = -acc[6 + s];         // M02 = -Sx
                A[(3 + r) * 9 + 6 + c] = -acc[12 + s]; A[(6 + r) * 9 + 3 + c] = -acc[12 + s]; // M12 = -Sy
                A[(6 + r) * 9 + 6 + c] = acc[18 + s];    // M22 = Sr
            }
    }
    __syncthreads();
    // keep a copy of the diagonal? Jacobi leaves eigenvalues on the diagonal of A.
    jacobi9(A, V, lane);
    if (lane == 0) {
        int m = 0;
        for (int i = 1; i < 9; ++i) if (A[i * 9 + i] < A[m * 9 + m]) m = i;
        double* e = eig + (size_t)prob * 96;
        for (int i = 0; i < 81; ++i) e[i] = V[i];
        for (int i = 0; i < 9; ++i) e[81 + i] = A[i * 9 + i];
        e[90] = (double)m;
        double Hh[9];
        for (int i = 0; i < 9; ++i) Hh[i] = V[i * 9 + m];
        double T1[9] = {t1.s, 0, -t1.s * t1.mx, 0, t1.s, -t1.s * t1.my, 0, 0, 1};
        double T2i[9] = {1.0 / t2.s, 0, t2.mx, 0, 1.0 / t2.s, t2.my, 0, 0, 1};
        double tmp[9], Hu[9];
        mat3_mul(Hh, T1, tmp);
        mat3_mul(T2i, tmp, Hu);
        double inv = 1.0 / (Hu[8] + 1e-8);
        double Hn[9];
        for (int i = 0; i < 9; ++i) { Hn[i] = Hu[i] * inv; Hout[(size_t)prob * 9 + i] = (float)Hn[i]; }
        for (int c = 0; c < 4; ++c) {
            double x, y;
            corner_xy(c, (double)w, (double)h, x, y);
            double qx = Hn[0] * x + Hn[1] * y + Hn[2], qy = Hn[3] * x + Hn[4] * y + Hn[5], qz = Hn[6] * x + Hn[7] * y + Hn[8];
            double sc = fabs(qz) > 1e-8 ? 1.0 / qz : 1.0;
            delta_hat[(size_t)prob * 8 + 2 * c] = (float)(qx * sc - x);
            delta_hat[(size_t)prob * 8 + 2 * c + 1] = (float)(qy * sc - y);
        }
    }
}

__global__ void __launch_bounds__(64) dlt_bwd_kernel(const float* __restrict__ pf, const int64_t* __restrict__ choice,
                                                     const double* __restrict__ eig, const float* __restrict__ g_delta,
                                                     const double* __restrict__ g_Hd, int P, int h, int w,
                                                     float* __restrict__ g_pf, int n, int j0, int det) {
    // n hypotheses per sample; this launch covers j0 + blockIdx.y.  det (deterministic mode: one launch per hypothesis, so the
    // hypotheses of a sample add to its field in stream order): repeated sample indices inside the wave are added up in point
    // order by the first of them and stored without atomics
    __shared__ double Gs[81];
    __shared__ double sc[8];   // g(1/s2), g m2x, g m2y
    __shared__ int d_idx[64 * DLT_MAXPTS];
    __shared__ float d_vx[64 * DLT_MAXPTS], d_vy[64 * DLT_MAXPTS];
    const int b = blockIdx.x, j = j0 + blockIdx.y, lane = threadIdx.x;
    const int prob = b * n + j;
    double x1[DLT_MAXPTS], y1[DLT_MAXPTS], x2[DLT_MAXPTS], y2[DLT_MAXPTS];
    int idx[DLT_MAXPTS];
    int cnt = load_points(pf, choice, b, j, n, P, h, w, lane, x1, y1, x2, y2, idx);
    Hartley t1 = hartley_stats(x1, y1, cnt, P);
    Hartley t2 = hartley_stats(x2, y2, cnt, P);
    const double* e = eig + (size_t)prob * 96;
    if (lane == 0) {
        int m = (int)e[90];
        double Hh[9];
        for (int i = 0; i < 9; ++i) Hh[i] = e[i * 9 + m];
        double T1[9] = {t1.s, 0, -t1.s * t1.mx, 0, t1.s, -t1.s * t1.my, 0, 0, 1};
        double T2i[9] = {1.0 / t2.s, 0, t2.mx, 0, 1.0 / t2.s, t2.my, 0, 0, 1};
        double HT1[9], Hu[9];
        mat3_mul(Hh, T1, HT1);
        mat3_mul(T2i, HT1, Hu);
        double den = Hu[8] + 1e-8, inv = 1.0 / den;
        double Hn[9];
        for (int i = 0; i < 9; ++i) Hn[i] = Hu[i] * inv;
        // g wrt Hn from the 4 corner transforms
        double gHn[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int c = 0; c < 4; ++c) {
            double x, y;
            corner_xy(c, (double)w, (double)h, x, y);
            double gx = (double)g_delta[(size_t)prob * 8 + 2 * c], gy = (double)g_delta[(size_t)prob * 8 + 2 * c + 1];
            double qx = Hn[0] * x + Hn[1] * y + Hn[2], qy = Hn[3] * x + Hn[4] * y + Hn[5], qz = Hn[6] * x + Hn[7] * y + Hn[8];
            if (fabs(qz) > 1e-8) {
                double iz = 1.0 / qz;
                double gqx = gx * iz, gqy = gy * iz, gqz = -(gx * qx + gy * qy) * iz * iz;
                gHn[0] += gqx * x; gHn[1] += gqx * y; gHn[2] += gqx;
                gHn[3] += gqy * x; gHn[4] += gqy * y; gHn[5] += gqy;
                gHn[6] += gqz * x; gHn[7] += gqz * y; gHn[8] += gqz;
            } else {
                gHn[0] += gx * x; gHn[1] += gx * y; gHn[2] += gx;
                gHn[3] += gy * x; gHn[4] += gy * y; gHn[5] += gy;
            }
        }
        // gradient that reaches the homography itself (hypothesis scoring, ransac_utils.py:76-128)
        if (g_Hd)
            for (int i = 0; i < 9; ++i) gHn[i] += g_Hd[(size_t)prob * 9 + i];
        // Hn = Hu / (Hu22 + eps)
        double gHu[9], dot = 0;
        for (int i = 0; i < 9; ++i) { gHu[i] = gHn[i] * inv; dot += gHn[i] * Hu[i]; }
        gHu[8] -= dot * inv * inv;
        // Hu = T2i * HT1 :  gT2i = gHu * HT1^T ; gHT1 = T2i^T gHu ; gHh = gHT1 * T1^T
        double gT2i[9], gHT1[9], gHh[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                gT2i[r * 3 + c] = gHu[r * 3] * HT1[c * 3] + gHu[r * 3 + 1] * HT1[c * 3 + 1] + gHu[r * 3 + 2] * HT1[c * 3 + 2];
                gHT1[r * 3 + c] = T2i[r] * gHu[c] + T2i[3 + r] * gHu[3 + c] + T2i[6 + r] * gHu[6 + c];
            }
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c)
                gHh[r * 3 + c] = gHT1[r * 3] * T1[c * 3] + gHT1[r * 3 + 1] * T1[c * 3 + 1] + gHT1[r * 3 + 2] * T1[c * 3 + 2];
        sc[0] = gT2i[0] + gT2i[4];   // g(1/s2)
        sc[1] = gT2i[2];             // g m2x
        sc[2] = gT2i[5];             // g m2y
        // eigenvector adjoint: G = sum_{i != m} c_i v_i v_m^T, c_i = (v_i . g)/(lam_m - lam_i); Gs = G + G^T
        double lm = e[81 + m];
        for (int i = 0; i < 81; ++i) Gs[i] = 0;
        for (int i = 0; i < 9; ++i) {
            if (i == m) continue;
            double d = 0;
            for (int k = 0; k < 9; ++k) d += e[k * 9 + i] * gHh[k];
            double ci = d / (lm - e[81 + i]);
            for (int r = 0; r < 9; ++r)
                for (int c = 0; c < 9; ++c) {
                    double t = ci * e[r * 9 + i] * e[c * 9 + m];
                    Gs[r * 9 + c] += t; Gs[c * 9 + r] += t;
                }
        }
    }
    __syncthreads();
    // per point: dL/du, dL/dv in normalised coordinates
    double gu[DLT_MAXPTS], gv[DLT_MAXPTS];
    double gs_acc = 0, gmx_acc = 0, gmy_acc = 0;
    for (int k = 0; k < cnt; ++k) {
        double a[3] = {t1.s * (x1[k] - t1.mx), t1.s * (y1[k] - t1.my), 1.0};
        double u = t2.s * (x2[k] - t2.mx), v = t2.s * (y2[k] - t2.my);
        double q02 = 0, q12 = 0, q22 = 0;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                double ar = a[r] * a[c];
                q02 += ar * Gs[r * 9 + 6 + c];
                q12 += ar * Gs[(3 + r) * 9 + 6 + c];
                q22 += ar * Gs[(6 + r) * 9 + 6 + c];
            }
        gu[k] = -q02 + u * q22;
        gv[k] = -q12 + v * q22;
        // u = s (x2 - mx)
        gs_acc += gu[k] * (x2[k] - t2.mx) + gv[k] * (y2[k] - t2.my);
        gmx_acc -= t2.s * gu[k];
        gmy_acc -= t2.s * gv[k];
    }
    gs_acc = wave_sum(gs_acc); gmx_acc = wave_sum(gmx_acc); gmy_acc = wave_sum(gmy_acc);
    double gs = gs_acc - sc[0] / (t2.s * t2.s);
    double gmx = gmx_acc + sc[1], gmy = gmy_acc + sc[2];
    // s = sqrt2 / (dbar + eps)
    double gdbar = -gs * t2.s / (t2.dbar + 1e-8);
    // dbar = mean ||p - m||: contributes to p_k and (through -sum) to m
    double gm_from_d_x = 0, gm_from_d_y = 0;
    double gpx[DLT_MAXPTS], gpy[DLT_MAXPTS];
    for (int k = 0; k < cnt; ++k) {
        double dx = x2[k] - t2.mx, dy = y2[k] - t2.my;
        double nrm = sqrt(dx * dx + dy * dy);
        double ux = nrm > 0 ? dx / nrm : 0.0, uy = nrm > 0 ? dy / nrm : 0.0;
        double tx = gdbar / P * ux, ty = gdbar / P * uy;
        gpx[k] = t2.s * gu[k] + tx;
        gpy[k] = t2.s * gv[k] + ty;
        gm_from_d_x -= tx; gm_from_d_y -= ty;
    }
    gmx += wave_sum(gm_from_d_x);
    gmy += wave_sum(gm_from_d_y);
    float* gpfx = g_pf + (size_t)b * 2 * h * w;
    float* gpfy = gpfx + (size_t)h * w;
    if (!det) {
        for (int k = 0; k < cnt; ++k) {
            atomicAdd(gpfx + idx[k], (float)(gpx[k] + gmx / P));
            atomicAdd(gpfy + idx[k], (float)(gpy[k] + gmy / P));
        }
        return;
    }
    for (int k = 0; k < DLT_MAXPTS; ++k) {                  // point p = lane + 64 k lives in slot p
        d_idx[k * 64 + lane] = k < cnt ? idx[k] : -1;
        d_vx[k * 64 + lane] = k < cnt ? (float)(gpx[k] + gmx / P) : 0.f;
        d_vy[k * 64 + lane] = k < cnt ? (float)(gpy[k] + gmy / P) : 0.f;
    }
    __syncthreads();
    const int np = P < 64 * DLT_MAXPTS ? P : 64 * DLT_MAXPTS;
    for (int k = 0; k < cnt; ++k) {
        const int me = k * 64 + lane, id = idx[k];
        bool first = true;
        float ax = 0.f, ay = 0.f;
        for (int q = 0; q < np; ++q)
            if (d_idx[q] == id) {
                if (q < me) first = false;
                ax += d_vx[q]; ay += d_vy[q];
            }
        if (first) { gpfx[id] += ax; gpfy[id] += ay; }
    }
}

// ---------------------------------------------------------------------------------------------
// DSAC scoring: grid (B, n), block 256; err = sum |H.coord - map|_1 ; then arg-min per sample
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) dsac_score_kernel(const float* __restrict__ pf, const float* __restrict__ Hd,
                                                         int h, int w, float* __restrict__ err) {
    __shared__ double part[4];
    const int b = blockIdx.x, j = blockIdx.y, n = gridDim.y;
    const float* Hm = Hd + (size_t)(b * n + j) * 9;
    float H0 = Hm[0], H1 = Hm[1], H2 = Hm[2], H3 = Hm[3], H4 = Hm[4], H5 = Hm[5], H6 = Hm[6], H7 = Hm[7], H8 = Hm[8];
    const float* pfx = pf + (size_t)b * 2 * h * w;
    const float* pfy = pfx + (size_t)h * w;
    double acc = 0;
    for (int i = threadIdx.x; i < h * w; i += 256) {
        float x = (float)(i % w), y = (float)(i / w);
        float qx = H0 * x + H1 * y + H2, qy = H3 * x + H4 * y + H5, qz = H6 * x + H7 * y + H8;
        float s = fabsf(qz) > 1e-8f ? 1.0f / qz : 1.0f;
        acc += (double)(fabsf(qx * s - (x + pfx[i])) + fabsf(qy * s - (y + pfy[i])));
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) err[b * n + j] = (float)(part[0] + part[1] + part[2] + part[3]);
}

// scores = softmax(-err) over the hypotheses of a sample (ransac_utils.py:126)
__global__ void dsac_softmax_kernel(const float* __restrict__ err, int B, int n, float* __restrict__ scores) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float mn = err[b * n];
    for (int j = 1; j < n; ++j) mn = fminf(mn, err[b * n + j]);
    float z = 0.f;
    for (int j = 0; j < n; ++j) z += expf(-(err[b * n + j] - mn));
    for (int j = 0; j < n; ++j) scores[b * n + j] = expf(-(err[b * n + j] - mn)) / z;
}

// g_scores -> g_err:  s = softmax(-e)  =>  de_j = -s_j (g_j - sum_k g_k s_k)
__global__ void dsac_softmax_bwd_kernel(const float* __restrict__ scores, const float* __restrict__ g_scores, int B, int n,
                                        float* __restrict__ g_err) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float dot = 0.f;
    for (int j = 0; j < n; ++j) dot += g_scores[b * n + j] * scores[b * n + j];
    for (int j = 0; j < n; ++j) g_err[b * n + j] = -scores[b * n + j] * (g_scores[b * n + j] - dot);
}

// adjoint of dsac_score_kernel: g_err[B,n] -> g_Hd[B*n,9] (double, overwritten) and g_pf[B,2,h,w] += (atomics: the n
// hypotheses of a sample and the DLT adjoint write the same field).  grid (B, n), block 256.
__global__ void __launch_bounds__(256) dsac_score_bwd_kernel(const float* __restrict__ pf, const float* __restrict__ Hd,
                                                             const float* __restrict__ g_err, int h, int w,
                                                             double* __restrict__ g_Hd, float* __restrict__ g_pf, int nhyp) {
    __shared__ double part[4][9];
    const int b = blockIdx.x, n = nhyp;
    for (int j = blockIdx.y; j < n; j += gridDim.y) {       // (deterministic mode: gridDim.y = 1 - one workgroup adds the hypotheses of a sample in order)
    const float* Hm = Hd + (size_t)(b * n + j) * 9;
    const float H0 = Hm[0], H1 = Hm[1], H2 = Hm[2], H3 = Hm[3], H4 = Hm[4], H5 = Hm[5], H6 = Hm[6], H7 = Hm[7], H8 = Hm[8];
    const float ge = g_err[b * n + j];
    const float* pfx = pf + (size_t)b * 2 * h * w;
    const float* pfy = pfx + (size_t)h * w;
    float* gx = g_pf + (size_t)b * 2 * h * w;
    float* gy = gx + (size_t)h * w;
    double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    // (pixel i is always handled by thread i % 256 of the sample's workgroup: with one workgroup per sample its adds to g_pf[i]
    //  follow the hypothesis order)
    for (int i = threadIdx.x; i < h * w; i += 256) {
        const float x = (float)(i % w), y = (float)(i / w);
        const float qx = H0 * x + H1 * y + H2, qy = H3 * x + H4 * y + H5, qz = H6 * x + H7 * y + H8;
        const bool ok = fabsf(qz) > 1e-8f;
        const float sc = ok ? 1.0f / qz : 1.0f;
        const float tx = qx * sc, ty = qy * sc;
        const float rx = tx - (x + pfx[i]), ry = ty - (y + pfy[i]);
        const float gtx = ge * (float)((rx > 0.f) - (rx < 0.f)), gty = ge * (float)((ry > 0.f) - (ry < 0.f));   // d|r| = sign(r), 0 at 0
        if (gtx != 0.f) atomicAdd(gx + i, -gtx);
        if (gty != 0.f) atomicAdd(gy + i, -gty);
        const double gqx = (double)(gtx * sc), gqy = (double)(gty * sc);
        const double gqz = ok ? -(double)(gtx * tx + gty * ty) * (double)sc : 0.0;
        s[0] += gqx * x; s[1] += gqx * y; s[2] += gqx;
        s[3] += gqy * x; s[4] += gqy * y; s[5] += gqy;
        s[6] += gqz * x; s[7] += gqz * y; s[8] += gqz;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) s[k] = wave_sum(s[k]);
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 9; ++k) part[threadIdx.x >> 6][k] = s[k];
    __syncthreads();
    if (threadIdx.x < 9)
        g_Hd[(size_t)(b * n + j) * 9 + threadIdx.x] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
    __syncthreads();
    }
}

__global__ void dsac_best_kernel(const float* __restrict__ err, int B, int n, int64_t* __restrict__ best) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int m = 0;
    float bv = err[b * n];
    for (int j = 1; j < n; ++j) {
        float v = err[b * n + j];
        if (v < bv) { bv = v; m = j; }      // first minimum == torch.argmax(softmax(-err)) tie rule
    }
    best[b] = m;
}

// ---------------------------------------------------------------------------------------------
extern "C" {

int bh_h4pt_fwd(const float* delta, int B, float W, float H, double* H64, float* H32, void* stream) {
    if (!delta || !H64 || B < 0) return BH_E_BADARG;
    if (B == 0) return BH_OK;
    hipLaunchKernelGGL(h4pt_fwd_kernel, dim3((B + 63) / 64), dim3(64), 0, bh_stream(stream), delta, B, W, H, H64, H32);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

int bh_h4pt_bwd(const float* delta, const double* H64, const double* gH, int B, float W, float H, float* gdelta,
                void* stream) {
    if (!delta || !H64 || !gH || !gdelta || B < 0) return BH_E_BADARG;
    if (B == 0) return BH_OK;
    hipLaunchKernelGGL(h4pt_bwd_kernel, dim3((B + 63) / 64), dim3(64), 0, bh_stream(stream), delta, H64, gH, B, W, H,
                       gdelta);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

int bh_dlt_fwd(const float* pf, const int64_t* choice, int B, int n, int P, int h, int w, float* Hdlt, float* delta_hat,
               double* eig, void* stream) {
    if (!pf || !choice || !Hdlt || !delta_hat || !eig || B < 0 || n < 1 || P < 4) return BH_E_BADARG;
    if (P > 64 * DLT_MAXPTS) return BH_E_UNSUPPORTED;
    if (B == 0) return BH_OK;
    hipLaunchKernelGGL(dlt_fwd_kernel, dim3(B, n), dim3(64), 0, bh_stream(stream), pf, choice, P, h, w, Hdlt, delta_hat,
                       eig);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

int bh_dlt_bwd(const float* pf, const int64_t* choice, const double* eig, const float* g_delta, const double* g_Hdlt, int B,
               int n, int P, int h, int w, float* g_pf, void* stream) {
    return bh_dlt_bwd_f(pf, choice, eig, g_delta, g_Hdlt, B, n, P, h, w, g_pf, 0, stream);
}

int bh_dlt_bwd_f(const float* pf, const int64_t* choice, const double* eig, const float* g_delta, const double* g_Hdlt, int B,
                 int n, int P, int h, int w, float* g_pf, int flags, void* stream) {
    if (!pf || !choice || !eig || !g_delta || !g_pf || B < 0 || n < 1 || P < 4) return BH_E_BADARG;
    if (P > 64 * DLT_MAXPTS) return BH_E_UNSUPPORTED;
    if (B == 0) return BH_OK;
    if (flags & BH_F_DETERMINISTIC) {
        for (int j = 0; j < n; ++j) {
            hipLaunchKernelGGL(dlt_bwd_kernel, dim3(B, 1), dim3(64), 0, bh_stream(stream), pf, choice, eig, g_delta, g_Hdlt, P, h, w,
                               g_pf, n, j, 1);
            BH_LAUNCH_CHECK();
        }
        return BH_OK;
    }
    hipLaunchKernelGGL(dlt_bwd_kernel, dim3(B, n), dim3(64), 0, bh_stream(stream), pf, choice, eig, g_delta, g_Hdlt, P, h, w,
                       g_pf, n, 0, 0);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

int bh_dsac_score(const float* pf, const float* Hdlt, int B, int n, int h, int w, float* err, int64_t* best,
                  void* stream) {
    if (!pf || !Hdlt || !err || B < 0 || n < 1) return BH_E_BADARG;
    if (B == 0) return BH_OK;
    hipLaunchKernelGGL(dsac_score_kernel, dim3(B, n), dim3(256), 0, bh_stream(stream), pf, Hdlt, h, w, err);
    BH_LAUNCH_CHECK();
    if (best) {
        hipLaunchKernelGGL(dsac_best_kernel, dim3((B + 63) / 64), dim3(64), 0, bh_stream(stream), err, B, n, best);
        BH_LAUNCH_CHECK();
    }
    return BH_OK;
}

int bh_dsac_scores_fwd(const float* err, int B, int n, float* scores, void* stream) {
    if (!err || !scores || B < 0 || n < 1) return BH_E_BADARG;
    if (B == 0) return BH_OK;
    hipLaunchKernelGGL(dsac_softmax_kernel, dim3((B + 63) / 64), dim3(64), 0, bh_stream(stream), err, B, n, scores);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

int bh_dsac_scores_bwd(const float* pf, const float* Hdlt, const float* scores, const float* g_scores, int B, int n, int h,
                       int w, float* g_err, double* g_Hdlt, float* g_pf, void* stream) {
    return bh_dsac_scores_bwd_f(pf, Hdlt, scores, g_scores, B, n, h, w, g_err, g_Hdlt, g_pf, 0, stream);
}

int bh_dsac_scores_bwd_f(const float* pf, const float* Hdlt, const float* scores, const float* g_scores, int B, int n, int h,
                         int w, float* g_err, double* g_Hdlt, float* g_pf, int flags, void* stream) {
    if (!pf || !Hdlt || !scores || !g_scores || !g_err || !g_Hdlt || !g_pf || B < 0 || n < 1) return BH_E_BADARG;
    if (B == 0) return BH_OK;
    hipLaunchKernelGGL(dsac_softmax_bwd_kernel, dim3((B + 63) / 64), dim3(64), 0, bh_stream(stream), scores, g_scores, B, n, g_err);
    BH_LAUNCH_CHECK();
    hipLaunchKernelGGL(dsac_score_bwd_kernel, dim3(B, (flags & BH_F_DETERMINISTIC) ? 1 : n), dim3(256), 0, bh_stream(stream), pf, Hdlt, g_err, h, w,
                       g_Hdlt, g_pf, n);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

}  // extern "C"
