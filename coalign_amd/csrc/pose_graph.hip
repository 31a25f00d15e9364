// Agent-object pose-graph optimisation for CoAlign's box alignment, gfx950, float64.
//
// Reference semantics (see include/coalign_amd.h): PoseGraphOptimization2D.optimize
// (opencood/models/sub_modules/pose_graph_optim.py:11-60) = g2o SparseOptimizer + OptimizationAlgorithmLevenberg over
// BlockSolverSE2(LinearSolverDenseSE2), on the graph built by box_align_v2.py:287-372: vertex 0 (ego) fixed, the other
// agents SE(2), one landmark (SE(2) or R^2) per box cluster, one edge per (agent, box) with a diagonal information matrix.
//
// One workgroup per graph (the caller batches the graphs of many frames into one launch).  The graph is bipartite --
// landmarks only connect to agents -- so the landmark block of the normal equations is block diagonal and the damped
// system (H + lambda I) dx = b reduces, by the Schur complement, to at most 3 x 8 pose unknowns:
//   linearise   thread per landmark: error, Jacobians A (agent side) / B (landmark side) of each of its edges,
//               H_ll, b_l, and per (landmark, agent) the sums  A^T W B,  A^T W A,  -A^T W e;
//   reduce      H_pp, b_p per agent (fixed summation order: deterministic), chi2, max diag(H) for lambda_0;
//   trial       thread per landmark (H_ll + lambda I)^-1; thread per entry of S = H_pp + lambda I - sum_l W (H_ll+lambda I)^-1 W^T
//               and of the reduced right-hand side; one wavefront factorises S (Cholesky, <= 24 x 24) and solves; thread per
//               landmark back-substitutes; the update is applied to a trial copy (SE2: X <- X * dx), chi2 re-evaluated;
//   decide      g2o's rule: rho = (chi2 - chi2') / (dx.(lambda dx + b) + 1e-3); accept -> lambda *= clamp(1-(2rho-1)^3, 1/3, 2/3);
//               reject -> lambda *= ni, ni *= 2; stop on rho == 0, ten rejections in a row, or max_iterations.
// The optimum does not depend on the elimination order; the trajectory follows the same damping schedule as g2o's.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxAgents = 8;
constexpr int kMaxLandmarks = 256;
constexpr int kDim = 3 * kMaxAgents;
// per-vertex workspace (doubles): trial estimate 3 | H_ll 6 | b_l 3 | (H_ll + lambda I)^-1 6 | dx 3 | per agent: A^T W B 9, A^T W A 6, -A^T W e 3
constexpr int kPerAgent = 18;
constexpr int kPerVertex = 21 + kMaxAgents * kPerAgent;
constexpr double kPi = 3.14159265358979323846;

struct GraphArgs {
    int n_graphs, max_iterations;
    const int *vertex_off, *edge_off, *n_agents;
    double *vertices;
    const int *kinds, *edge_agent, *edge_landmark;
    const double *edge_meas, *edge_info;
    double *stats, *ws;
};

__device__ __forceinline__ double normalize_theta(double t) {    // g2o misc.h
    if (t >= -kPi && t < kPi) return t;
    const double m = floor(t / (2 * kPi));
    t = t - m * 2 * kPi;
    if (t >= kPi) t -= 2 * kPi;
    if (t < -kPi) t += 2 * kPi;
    return t;
}

// error of one edge: SE2 landmark  e = toVector(M^-1 * (X1^-1 * X2));  point landmark  e = X1^-1 * l - m
__device__ __forceinline__ void edge_error(const double *a, const double *l, bool se2, const double *m, double *e) {
    const double c1 = cos(a[2]), s1 = sin(a[2]);
    const double dx = l[0] - a[0], dy = l[1] - a[1];
    const double rx = c1 * dx + s1 * dy - m[0], ry = -s1 * dx + c1 * dy - m[1];
    const double cm = se2 ? cos(m[2]) : 1.0, sm = se2 ? sin(m[2]) : 0.0;
    e[0] = cm * rx + sm * ry;
    e[1] = -sm * rx + cm * ry;
    e[2] = se2 ? normalize_theta(l[2] - a[2] - m[2]) : 0.0;
}

// Jacobians w.r.t. the local updates  X1 <- X1 * d1  and  X2 <- X2 * d2 (or l <- l + d2), row-major 3 x 3
__device__ __forceinline__ void edge_jacobians(const double *a, const double *l, bool se2, const double *m, double *A, double *B) {
    const double c1 = cos(a[2]), s1 = sin(a[2]);
    const double cm = se2 ? cos(m[2]) : 1.0, sm = se2 ? sin(m[2]) : 0.0;
    const double dx = l[0] - a[0], dy = l[1] - a[1];
    // d/dtheta1 of R1^T d  = [[-s, c], [-c, -s]] d
    const double gx = -s1 * dx + c1 * dy, gy = -c1 * dx - s1 * dy;
    A[0] = -cm; A[1] = -sm; A[2] = cm * gx + sm * gy;
    A[3] = sm;  A[4] = -cm; A[5] = -sm * gx + cm * gy;
    A[6] = 0;   A[7] = 0;   A[8] = se2 ? -1.0 : 0.0;
    // Rm^T R1^T (R2 for an SE2 landmark)
    double r00 = cm * c1 - sm * s1, r01 = cm * s1 + sm * c1;     // Rm^T R1^T = R(-(thm + th1)) -> [[C, S], [-S, C]]
    double q00 = r00, q01 = r01, q10 = -r01, q11 = r00;
    if (se2) {
        const double c2 = cos(l[2]), s2 = sin(l[2]);
        const double t00 = q00 * c2 + q01 * s2, t01 = -q00 * s2 + q01 * c2;
        const double t10 = q10 * c2 + q11 * s2, t11 = -q10 * s2 + q11 * c2;
        q00 = t00; q01 = t01; q10 = t10; q11 = t11;
    }
    B[0] = q00; B[1] = q01; B[2] = 0;
    B[3] = q10; B[4] = q11; B[5] = 0;
    B[6] = 0;   B[7] = 0;   B[8] = se2 ? 1.0 : 0.0;
}

__device__ __forceinline__ double block_sum(double v, double *scratch) {     // fixed tree: deterministic
    const int t = threadIdx.x;
    scratch[t] = v;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1) {
        if (t < s) scratch[t] += scratch[t + s];
        __syncthreads();
    }
    const double r = scratch[0];
    __syncthreads();
    return r;
}

__device__ __forceinline__ double block_max(double v, double *scratch) {
    const int t = threadIdx.x;
    scratch[t] = v;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1) {
        if (t < s) scratch[t] = fmax(scratch[t], scratch[t + s]);
        __syncthreads();
    }
    const double r = scratch[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(kThreads) void pose_graph_kernel(const GraphArgs g) {
    __shared__ double S[kDim][kDim + 1], rhs[kDim], dp[kDim], Hpp[kMaxAgents][6], bp[kMaxAgents][3], pose[kMaxAgents][3], trial[kMaxAgents][3];
    __shared__ double scratch[kThreads];
    __shared__ int lstart[kMaxLandmarks], lend[kMaxLandmarks], col_of[kMaxAgents];
    __shared__ int solve_ok, bad_graph;
    const int graph = blockIdx.x, t = threadIdx.x;
    const int v0 = g.vertex_off[graph], V = g.vertex_off[graph + 1] - v0;
    const int e0 = g.edge_off[graph], E = g.edge_off[graph + 1] - e0;
    const int NA = g.n_agents[graph], NL = V - NA;
    double *stats = g.stats + (size_t)graph * 4;
    double *est = g.vertices + (size_t)v0 * 3;
    const int *kinds = g.kinds + v0;
    const int *ea = g.edge_agent + e0, *el = g.edge_landmark + e0;
    const double *em = g.edge_meas + (size_t)e0 * 3, *ew = g.edge_info + (size_t)e0 * 3;
    double *ws = g.ws + (size_t)v0 * kPerVertex;
    auto W = [&](int l) { return ws + (size_t)(NA + l) * kPerVertex; };    // landmark l's workspace row

    // ---- structure: landmark edge ranges (edges are grouped by landmark), free-agent columns, validation
    if (t == 0) { bad_graph = (NA < 1 || NA > kMaxAgents || NL < 0 || NL > kMaxLandmarks) ? 1 : 0; }
    for (int l = t; l < kMaxLandmarks; l += kThreads) { lstart[l] = 0; lend[l] = 0; }
    __syncthreads();
    if (!bad_graph) {
        for (int k = t; k < E; k += kThreads) {
            const int a = ea[k], l = el[k] - NA;
            if (a < 0 || a >= NA || l < 0 || l >= NL) { bad_graph = 1; continue; }
            if (k == 0 || el[k - 1] != el[k]) lstart[l] = k;
            if (k == E - 1 || el[k + 1] != el[k]) lend[l] = k + 1;
        }
    }
    __syncthreads();
    if (!bad_graph) {           // a landmark whose edges are not contiguous would have been overwritten: count check
        int covered = 0;
        for (int l = t; l < NL; l += kThreads) covered += lend[l] - lstart[l];
        if ((int)block_sum((double)covered, scratch) != E && t == 0) bad_graph = 1;
    }
    __syncthreads();
    if (bad_graph) {
        if (t == 0) { stats[0] = -1; stats[1] = stats[2] = stats[3] = 0; }
        return;
    }
    if (t == 0) {
        int c = 0;
        for (int a = 0; a < NA; ++a) col_of[a] = -1;
        for (int a = 0; a < NA; ++a) {          // free agents with at least one edge get columns (g2o's active set)
            bool used = false;
            for (int k = 0; k < E && !used; ++k) used = ea[k] == a;
            if (kinds[a] != 0 && used) { col_of[a] = c; c += 3; }
        }
        scratch[0] = c;
    }
    __syncthreads();
    const int n = (int)scratch[0];
    __syncthreads();
    for (int i = t; i < NA * 3; i += kThreads) pose[i / 3][i % 3] = est[i];
    __syncthreads();
    const bool my_lm = t < NL && lend[t] > lstart[t];           // this thread's landmark is active
    const bool my_se2 = my_lm && kinds[NA + t] == 1;
    double lm[3] = {0, 0, 0};
    if (my_lm) { lm[0] = est[(NA + t) * 3]; lm[1] = est[(NA + t) * 3 + 1]; lm[2] = my_se2 ? est[(NA + t) * 3 + 2] : 0.0; }

    double lambda = 0, chi = 0, chi0 = 0;
    int it = 0;
    for (; it < g.max_iterations; ++it) {
        // ---------------------------------------------------------------------------------------- linearise at (pose, lm)
        double Hll[6] = {0, 0, 0, 0, 0, 0}, bl[3] = {0, 0, 0}, my_chi = 0;
        if (my_lm) {
            double *row = W(t);
            for (int a = 0; a < NA; ++a)
                for (int q = 0; q < kPerAgent; ++q) row[21 + a * kPerAgent + q] = 0;
            for (int k = lstart[t]; k < lend[t]; ++k) {
                const int a = ea[k];
                double e[3], A[9], B[9];
                const double m[3] = {em[k * 3], em[k * 3 + 1], em[k * 3 + 2]}, w[3] = {ew[k * 3], ew[k * 3 + 1], ew[k * 3 + 2]};
                edge_error(pose[a], lm, my_se2, m, e);
                edge_jacobians(pose[a], lm, my_se2, m, A, B);
                my_chi += e[0] * e[0] * w[0] + e[1] * e[1] * w[1] + e[2] * e[2] * w[2];
                double *pa = row + 21 + a * kPerAgent;
                int u = 0;
                for (int i = 0; i < 3; ++i)
                    for (int j = i; j < 3; ++j, ++u) {
                        double hb = 0, ha = 0;
                        for (int r = 0; r < 3; ++r) { hb += B[r * 3 + i] * w[r] * B[r * 3 + j]; ha += A[r * 3 + i] * w[r] * A[r * 3 + j]; }
                        Hll[u] += hb;
                        pa[9 + u] += ha;
                    }
                for (int i = 0; i < 3; ++i) {
                    double gb = 0, ga = 0;
                    for (int r = 0; r < 3; ++r) { gb += B[r * 3 + i] * w[r] * e[r]; ga += A[r * 3 + i] * w[r] * e[r]; }
                    bl[i] -= gb;
                    pa[15 + i] -= ga;
                    for (int j = 0; j < 3; ++j) {
                        double x = 0;
                        for (int r = 0; r < 3; ++r) x += A[r * 3 + i] * w[r] * B[r * 3 + j];
                        pa[i * 3 + j] += x;
                    }
                }
            }
        }
        chi = block_sum(my_chi, scratch);
        // H_pp, b_p: thread per (agent, component), landmarks in index order
        if (t < NA * 9) {
            const int a = t / 9, q = t % 9;
            double s = 0;
            for (int l = 0; l < NL; ++l)
                if (lend[l] > lstart[l]) s += W(l)[21 + a * kPerAgent + 9 + q];
            if (q < 6) Hpp[a][q] = s; else bp[a][q - 6] = s;
        }
        __syncthreads();
        if (it == 0) {
            double d = my_lm ? fmax(Hll[0], fmax(Hll[3], my_se2 ? Hll[5] : 0.0)) : 0.0;
            if (t < NA && col_of[t] >= 0) d = fmax(d, fmax(Hpp[t][0], fmax(Hpp[t][3], Hpp[t][5])));
            lambda = 1e-5 * block_max(d, scratch);
            chi0 = chi;
        }
        // ---------------------------------------------------------------------------------------- damped trials
        double ni = 2, rho = 0;
        int qmax = 0;
        do {
            // (H_ll + lambda I)^-1, symmetric 3 x 3 by cofactors (rows of a point landmark: the unused 3rd dimension decouples)
            double Mi[6] = {0, 0, 0, 0, 0, 0};
            bool inv_ok = true;
            if (my_lm) {
                const double a00 = Hll[0] + lambda, a01 = Hll[1], a02 = Hll[2], a11 = Hll[3] + lambda, a12 = Hll[4], a22 = Hll[5] + lambda;
                const double c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
                const double det = a00 * c00 + a01 * c01 + a02 * c02;
                inv_ok = det > 0 && isfinite(det);
                const double id = 1.0 / det;
                Mi[0] = c00 * id; Mi[1] = c01 * id; Mi[2] = c02 * id;
                Mi[3] = (a00 * a22 - a02 * a02) * id; Mi[4] = (a01 * a02 - a00 * a12) * id; Mi[5] = (a00 * a11 - a01 * a01) * id;
                double *row = W(t);
                for (int q = 0; q < 6; ++q) row[12 + q] = Mi[q];
                for (int q = 0; q < 3; ++q) row[9 + q] = bl[q];
            }
            if (t == 0) solve_ok = 1;
            __syncthreads();
            if (!inv_ok) solve_ok = 0;
            // reduced system: S[(a,i)][(a',j)], rhs[(a,i)]
            for (int idx = t; idx < n * (n + 1); idx += kThreads) {
                const int ri = idx / (n + 1), cj = idx % (n + 1);
                int a = 0, b = 0;
                for (int q = 0; q < NA; ++q) { if (col_of[q] >= 0 && col_of[q] <= ri) a = q; if (col_of[q] >= 0 && col_of[q] <= cj) b = q; }
                const int i = ri - col_of[a];
                double s;
                if (cj == n) {
                    s = bp[a][i];
                    for (int l = 0; l < NL; ++l) {
                        if (lend[l] <= lstart[l]) continue;
                        const double *row = W(l), *wa = row + 21 + a * kPerAgent + i * 3, *mi = row + 12, *b3 = row + 9;
                        const double y0 = mi[0] * b3[0] + mi[1] * b3[1] + mi[2] * b3[2], y1 = mi[1] * b3[0] + mi[3] * b3[1] + mi[4] * b3[2],
                                     y2 = mi[2] * b3[0] + mi[4] * b3[1] + mi[5] * b3[2];
                        s -= wa[0] * y0 + wa[1] * y1 + wa[2] * y2;
                    }
                    rhs[ri] = s;
                } else {
                    const int j = cj - col_of[b];
                    s = 0;
                    if (a == b) {
                        const int lo = i < j ? i : j, hi = i < j ? j : i;
                        s = Hpp[a][lo == 0 ? hi : (lo == 1 ? 2 + hi : 5)] + (i == j ? lambda : 0.0);
                    }
                    for (int l = 0; l < NL; ++l) {
                        if (lend[l] <= lstart[l]) continue;
                        const double *row = W(l), *wa = row + 21 + a * kPerAgent + i * 3, *wb = row + 21 + b * kPerAgent + j * 3, *mi = row + 12;
                        const double y0 = mi[0] * wb[0] + mi[1] * wb[1] + mi[2] * wb[2], y1 = mi[1] * wb[0] + mi[3] * wb[1] + mi[4] * wb[2],
                                     y2 = mi[2] * wb[0] + mi[4] * wb[1] + mi[5] * wb[2];
                        s -= wa[0] * y0 + wa[1] * y1 + wa[2] * y2;
                    }
                    S[ri][cj] = s;
                }
            }
            __syncthreads();
            // Cholesky S = L L^T in place (lower), forward / backward substitution: first wavefront, one row per lane
            if (t < 64) {
                for (int k = 0; k < n; ++k) {
                    const double d = S[k][k];
                    if (!(d > 0) || !isfinite(d)) { if (t == 0) solve_ok = 0; break; }
                    const double r = sqrt(d);
                    if (t == k) S[k][k] = r;
                    if (t > k && t < n) S[t][k] = S[t][k] / r;
                    coalign::wave_lds_sync();
                    if (t > k && t < n)
                        for (int j = k + 1; j <= t; ++j) S[t][j] -= S[t][k] * S[j][k];
                    coalign::wave_lds_sync();
                }
                coalign::wave_lds_sync();
                if (solve_ok) {
                    if (t == 0) {
                        for (int i = 0; i < n; ++i) {
                            double s = rhs[i];
                            for (int j = 0; j < i; ++j) s -= S[i][j] * dp[j];
                            dp[i] = s / S[i][i];
                        }
                        for (int i = n - 1; i >= 0; --i) {
                            double s = dp[i];
                            for (int j = i + 1; j < n; ++j) s -= S[j][i] * dp[j];
                            dp[i] = s / S[i][i];
                        }
                    }
                }
            }
            __syncthreads();
            const bool ok = solve_ok != 0;
            // back-substitution for the landmarks, trial update, scale = dx . (lambda dx + b)
            double dl[3] = {0, 0, 0}, scale_part = 0, tl[3] = {lm[0], lm[1], lm[2]};
            if (ok && my_lm) {
                const double *row = W(t);
                double r[3] = {bl[0], bl[1], bl[2]};
                for (int a = 0; a < NA; ++a) {
                    if (col_of[a] < 0) continue;
                    const double *wa = row + 21 + a * kPerAgent, *d = dp + col_of[a];
                    for (int j = 0; j < 3; ++j) r[j] -= wa[j] * d[0] + wa[3 + j] * d[1] + wa[6 + j] * d[2];
                }
                dl[0] = Mi[0] * r[0] + Mi[1] * r[1] + Mi[2] * r[2];
                dl[1] = Mi[1] * r[0] + Mi[3] * r[1] + Mi[4] * r[2];
                dl[2] = my_se2 ? Mi[2] * r[0] + Mi[4] * r[1] + Mi[5] * r[2] : 0.0;
                for (int j = 0; j < 3; ++j) scale_part += dl[j] * (lambda * dl[j] + bl[j]);
                if (my_se2) {
                    const double c = cos(lm[2]), s = sin(lm[2]);
                    tl[0] = lm[0] + c * dl[0] - s * dl[1];
                    tl[1] = lm[1] + s * dl[0] + c * dl[1];
                    tl[2] = normalize_theta(lm[2] + dl[2]);
                } else {
                    tl[0] = lm[0] + dl[0];
                    tl[1] = lm[1] + dl[1];
                }
            }
            if (t < NA) {
                for (int j = 0; j < 3; ++j) trial[t][j] = pose[t][j];
                if (ok && col_of[t] >= 0) {
                    const double *d = dp + col_of[t];
                    const double c = cos(pose[t][2]), s = sin(pose[t][2]);
                    trial[t][0] = pose[t][0] + c * d[0] - s * d[1];
                    trial[t][1] = pose[t][1] + s * d[0] + c * d[1];
                    trial[t][2] = normalize_theta(pose[t][2] + d[2]);
                    for (int j = 0; j < 3; ++j) scale_part += d[j] * (lambda * d[j] + bp[t][j]);
                }
            }
            __syncthreads();
            double new_part = 0;
            if (ok && my_lm)
                for (int k = lstart[t]; k < lend[t]; ++k) {
                    double e[3];
                    const double m[3] = {em[k * 3], em[k * 3 + 1], em[k * 3 + 2]};
                    edge_error(trial[ea[k]], tl, my_se2, m, e);
                    new_part += e[0] * e[0] * ew[k * 3] + e[1] * e[1] * ew[k * 3 + 1] + e[2] * e[2] * ew[k * 3 + 2];
                }
            const double new_chi = block_sum(new_part, scratch);
            const double scale = ok ? block_sum(scale_part, scratch) + 1e-3 : 1.0;
            rho = ok ? (chi - new_chi) / scale : -1.0;
            if (rho > 0 && isfinite(new_chi) && ok) {
                double alpha = 1.0 - (2 * rho - 1) * (2 * rho - 1) * (2 * rho - 1);
                alpha = fmin(alpha, 2.0 / 3.0);
                lambda *= fmax(1.0 / 3.0, alpha);
                ni = 2;
                chi = new_chi;
                for (int j = 0; j < 3; ++j) lm[j] = tl[j];
                if (t < NA)
                    for (int j = 0; j < 3; ++j) pose[t][j] = trial[t][j];
            } else {
                lambda *= ni;
                ni *= 2;
                if (!isfinite(lambda)) { ++qmax; break; }
            }
            ++qmax;
            __syncthreads();
        } while (rho < 0 && qmax < 10);
        if (qmax == 10 || rho == 0 || !isfinite(lambda)) { ++it; break; }
    }
    __syncthreads();
    for (int i = t; i < NA * 3; i += kThreads)
        if (kinds[i / 3] != 0) est[i] = pose[i / 3][i % 3];
    if (my_lm) {
        est[(NA + t) * 3] = lm[0];
        est[(NA + t) * 3 + 1] = lm[1];
        if (my_se2) est[(NA + t) * 3 + 2] = lm[2];
    }
    if (t == 0) { stats[0] = it; stats[1] = chi0; stats[2] = chi; stats[3] = lambda; }
}

}  // namespace

extern "C" size_t coalign_pose_graph_workspace_bytes(int total_vertices) {
    return total_vertices < 0 ? 0 : (size_t)(total_vertices + 1) * kPerVertex * sizeof(double);
}

extern "C" int coalign_pose_graph_optimize(int n_graphs, const int32_t *vertex_offsets, const int32_t *edge_offsets,
                                           const int32_t *n_agents, int total_vertices, double *vertices, const int32_t *kinds,
                                           const int32_t *edge_agent, const int32_t *edge_landmark, const double *edge_meas,
                                           const double *edge_info, int max_iterations, double *stats, void *workspace,
                                           size_t workspace_bytes, void *stream) {
    using namespace coalign;
    if (n_graphs < 0 || total_vertices < 0 || max_iterations < 0) return COALIGN_ERR_BAD_SHAPE;
    if (n_graphs == 0) return COALIGN_OK;
    if (!vertex_offsets || !edge_offsets || !n_agents || !vertices || !kinds || !stats || !workspace) return COALIGN_ERR_NULL_POINTER;
    if (workspace_bytes < coalign_pose_graph_workspace_bytes(total_vertices)) return COALIGN_ERR_WORKSPACE;
    GraphArgs g{n_graphs, max_iterations, vertex_offsets, edge_offsets, n_agents, vertices, kinds, edge_agent, edge_landmark,
                edge_meas, edge_info, stats, static_cast<double *>(workspace)};
    hipLaunchKernelGGL(pose_graph_kernel, dim3(n_graphs), dim3(kThreads), 0, static_cast<hipStream_t>(stream), g);
    return check_launch();
}
