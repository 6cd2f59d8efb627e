/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the point-cloud projection path.
 *
 * Scalar fp32 restatement of the reference's *literal* arithmetic (SURVEY.md
 * Appendix A), one function per reference stage, each citing the reference
 * file:line (relative to /root/reference/code) it follows.  Nothing in the
 * product path (2dimageto3dmodel_amd/) may link, load or call this file; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only
 * as the checker.
 *
 * Pinned against golden vectors produced by executing the reference's own
 * Python on CPU (oracle/gen_golden_p.py -> tests/golden/p_*.npz):
 *   - orc_transform   : bit-exact  (0 mismatching coordinates)
 *   - bins / in-bounds: exact
 *   - volume          : bit-exact on the goldens (same accumulation order)
 *   - smooth, termination, projection, loss: <= few ulp (libm logf/expf vs
 *     torch's SLEEF; conv summation order), checked at 1e-6 relative
 *   - gradients       : analytic, checked against the reference's autograd
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (oracle/Makefile).
 * -ffp-contract=off is REQUIRED: torch-CPU rounds every elementwise op
 * separately; one fused multiply-add changes bin indices.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FOV 1.875f
#define CAM_DIST 2.0f

/* ---- P1: quaternions/points_quaternions.py:41-81, quaternions/operations.py:68-97,120-136 ---- */
static void hamilton(const float a[4], const float b[4], float r[4])
{
    /* operations.py:82-85 -- left-to-right, every op rounded */
    r[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    r[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    r[2] = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
    r[3] = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
}

void orc_normalize_quat(const float q[4], float qn[4])
{
    /* F.normalize (points_quaternions.py:52-55): q / max(||q||_2, 1e-12); the
     * sequential sum ((a0^2+a1^2)+a2^2)+a3^2 reproduces torch-CPU bit for bit */
    float s = q[0] * q[0];
    s = s + q[1] * q[1];
    s = s + q[2] * q[2];
    s = s + q[3] * q[3];
    float n = sqrtf(s);
    if (n < 1e-12f) n = 1e-12f;
    for (int i = 0; i < 4; ++i) qn[i] = q[i] / n;
}

/* P1+P2: camera/coordinate_system_transformation.py:20-39.  cam is [B,N,3] in (z,y,x) order. */
void orc_transform(const float *p, const float *q, int B, int N, float *cam)
{
    for (int b = 0; b < B; ++b) {
        float qn[4], qs[4];
        orc_normalize_quat(q + 4 * b, qn);
        /* operations.py:131-136: q * (1,-1,-1,-1) */
        qs[0] = qn[0] * 1.0f; qs[1] = qn[1] * -1.0f; qs[2] = qn[2] * -1.0f; qs[3] = qn[3] * -1.0f;
        for (int n = 0; n < N; ++n) {
            const float *pp = p + ((size_t)b * N + n) * 3;
            float p4[4] = {0.0f, pp[0], pp[1], pp[2]}; /* points_quaternions.py:33 */
            float t[4], r[4];
            hamilton(qn, p4, t);  /* points_quaternions.py:72-75 */
            hamilton(t, qs, r);
            float z = r[1], y = r[2], x = r[3];      /* cam:25 unbind -> z,y,x */
            float den = z + CAM_DIST;
            float xo = x * FOV / den;                 /* cam:33 */
            float yo = y * FOV / den;                 /* cam:34 */
            float *o = cam + ((size_t)b * N + n) * 3;
            o[0] = z; o[1] = yo; o[2] = xo;           /* cam:36-39 */
        }
    }
}

/* P3 helpers: utils/trilinear_interpolation.py:17-35 */
static int in_bounds(const float c[3])
{
    const float hi = (float)(0.5 - 1e-6), lo = (float)(-0.5 + 1e-6); /* tri:24, scalar cast to f32 */
    return c[0] < hi && c[0] > lo && c[1] < hi && c[1] > lo && c[2] < hi && c[2] > lo;
}

/* bins[B,N,4] = (inb, f0, f1, f2) : the index-exactness contract */
void orc_bins(const float *cam, int B, int N, int S, int32_t *bins)
{
    const float sm1 = (float)S - 1.0f; /* tri:34 (voxel_size - 1) on an fp32 tensor */
    for (size_t i = 0; i < (size_t)B * N; ++i) {
        const float *c = cam + 3 * i;
        bins[4 * i + 0] = in_bounds(c);
        for (int a = 0; a < 3; ++a) bins[4 * i + 1 + a] = (int32_t)floorf(sm1 * (c[a] + 0.5f));
    }
}

/* P3: trilinear_interpolation (tri:62-74).  fixed_weights=0 -> literal w0 = 1-g-floor(g) (defect D3).
 * Reference order: 8 corner volumes, each accumulated sequentially in point order
 * (index_put_ accumulate on CPU is serial), then stack().sum(0), then clamp. */
void orc_splat(const float *cam, int B, int N, int S, int fixed_weights, float *V)
{
    const float sm1 = (float)S - 1.0f;
    size_t vol = (size_t)S * S * S;
    float *corner = (float *)malloc(sizeof(float) * vol);
    for (int b = 0; b < B; ++b) {
        float *Vb = V + (size_t)b * vol;
        memset(Vb, 0, sizeof(float) * vol);
        for (int c8 = 0; c8 < 8; ++c8) {
            int ci = (c8 >> 2) & 1, cj = (c8 >> 1) & 1, ck = c8 & 1; /* tri:68-70 loop order i,j,k */
            memset(corner, 0, sizeof(float) * vol);
            for (int n = 0; n < N; ++n) {
                const float *c = cam + ((size_t)b * N + n) * 3;
                if (!in_bounds(c)) continue;
                float g[3], f[3], w[3];
                int sel[3] = {ci, cj, ck};
                for (int a = 0; a < 3; ++a) {
                    g[a] = sm1 * (c[a] + 0.5f);
                    f[a] = floorf(g[a]);
                    if (sel[a]) w[a] = g[a] - f[a];                       /* tri:66 second entry */
                    else w[a] = fixed_weights ? 1.0f - (g[a] - f[a]) : (1.0f - g[a]) - f[a]; /* tri:66 */
                }
                float upd = w[0] * w[1] * w[2];                            /* tri:40-41 */
                size_t idx = ((size_t)((int)f[0] + ci) * S + ((int)f[1] + cj)) * S + ((int)f[2] + ck);
                corner[idx] += upd;                                        /* tri:58 */
            }
            for (size_t i = 0; i < vol; ++i) Vb[i] = Vb[i] + corner[i];   /* tri:74 stack().sum(0) */
        }
        for (size_t i = 0; i < vol; ++i) {                                 /* tri:74 clamp */
            float v = Vb[i];
            Vb[i] = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
        }
    }
    free(corner);
}

/* P3 raw (pre-clamp) sum, needed by the analytic backward (clamp mask) */
static void splat_raw(const float *cam, int b, int N, int S, int fixed_weights, float *Vb)
{
    const float sm1 = (float)S - 1.0f;
    size_t vol = (size_t)S * S * S;
    memset(Vb, 0, sizeof(float) * vol);
    for (int n = 0; n < N; ++n) {
        const float *c = cam + ((size_t)b * N + n) * 3;
        if (!in_bounds(c)) continue;
        float g[3], f[3], w[3][2];
        for (int a = 0; a < 3; ++a) {
            g[a] = sm1 * (c[a] + 0.5f);
            f[a] = floorf(g[a]);
            w[a][1] = g[a] - f[a];
            w[a][0] = fixed_weights ? 1.0f - (g[a] - f[a]) : (1.0f - g[a]) - f[a];
        }
        for (int c8 = 0; c8 < 8; ++c8) {
            int ci = (c8 >> 2) & 1, cj = (c8 >> 1) & 1, ck = c8 & 1;
            size_t idx = ((size_t)((int)f[0] + ci) * S + ((int)f[1] + cj)) * S + ((int)f[2] + ck);
            Vb[idx] += w[0][ci] * w[1][cj] * w[2][ck];
        }
    }
}

/* P4 taps: utils/smooth_voxels.py:24-31. literal=1 -> exp(+x^2/(2 sigma^2)) (defect D4) */
void orc_taps(float sigma, int ksize, int literal, float *taps)
{
    /* a,b = (-k//2, k//2); x = arange(a+1, b+1)  -> for k=21: -10..10 */
    int a = -((ksize + 1) / 2); /* python floor division: -21//2 = -11, -20//2 = -10 */
    int bb = ksize / 2;
    int n = bb - a; /* arange(a+1, b+1) has b-a entries */
    float den = 2.0f * (sigma * sigma);
    float sum = 0.0f;
    for (int i = 0; i < n; ++i) {
        float x = (float)(a + 1 + i);
        float e = (x * x) / den;
        taps[i] = expf(literal ? e : -e);
    }
    for (int i = 0; i < n; ++i) sum += taps[i];
    for (int i = 0; i < n; ++i) taps[i] = taps[i] / sum;
}

/* P4: smooth (sm:44-84).  axis_mask bit0=depth(z) bit1=y bit2=x.  literal_overwrite=1 reproduces
 * defect D5 (each conv reads the ORIGINAL volume; only the last kernel in the list -- depth -- survives):
 * with the reference's kernel list [x, y, depth] that is identical to axis_mask=1.
 * chained mode (literal_overwrite=0) applies x, then y, then depth in sequence.
 * has_scale: multiply by scale[b] then clamp(0,1) (sm:80-82).  zero padding ntaps/2. */
static void conv_axis(const float *in, float *out, int S, const float *taps, int ntaps, int axis)
{
    int half = ntaps / 2;
    size_t stride = axis == 0 ? (size_t)S * S : (axis == 1 ? (size_t)S : 1);
    for (int z = 0; z < S; ++z)
        for (int y = 0; y < S; ++y)
            for (int x = 0; x < S; ++x) {
                int pos = axis == 0 ? z : (axis == 1 ? y : x);
                size_t base = ((size_t)z * S + y) * S + x;
                float acc = 0.0f;
                for (int t = 0; t < ntaps; ++t) {
                    int q = pos + t - half;
                    if (q < 0 || q >= S) continue;
                    acc += taps[t] * in[base + (ptrdiff_t)(q - pos) * (ptrdiff_t)stride];
                }
                out[base] = acc;
            }
}

void orc_smooth(const float *V, int B, int S, const float *taps, int ntaps, int axis_mask,
                const float *scale, float *out)
{
    size_t vol = (size_t)S * S * S;
    float *tmp = (float *)malloc(sizeof(float) * vol);
    for (int b = 0; b < B; ++b) {
        const float *cur = V + (size_t)b * vol;
        float *ob = out + (size_t)b * vol;
        int first = 1;
        /* chained order: x (bit2), y (bit1), depth (bit0) -- sm:66 iterates [first(x), second(y), third(depth)] */
        int order[3] = {2, 1, 0};
        for (int oi = 0; oi < 3; ++oi) {
            int axis = order[oi];
            if (!(axis_mask & (1 << axis))) continue;
            if (first) { conv_axis(cur, ob, S, taps, ntaps, axis); first = 0; }
            else { memcpy(tmp, ob, sizeof(float) * vol); conv_axis(tmp, ob, S, taps, ntaps, axis); }
        }
        if (first) memcpy(ob, cur, sizeof(float) * vol);
        if (scale) {
            float s = scale[b];
            for (size_t i = 0; i < vol; ++i) {
                float v = ob[i] * s;
                ob[i] = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
            }
        }
    }
    free(tmp);
}

/* P5: termination_probs (elf:18-56). in [B,D,H,W] -> out [B,D+1,H,W].  cumsum accumulates in
 * double and rounds each prefix to float, as ATen's CPU cumsum does for float tensors. */
void orc_termination(const float *V, int B, int D, int H, int W, float eps_d, float *T)
{
    const float eps = eps_d, hi = (float)(1.0 - (double)eps_d);
    size_t hw = (size_t)H * W;
    for (int b = 0; b < B; ++b)
        for (size_t r = 0; r < hw; ++r) {
            const float *col = V + (size_t)b * D * hw + r;
            float *out = T + (size_t)b * (D + 1) * hw + r;
            double acc = 0.0;
            float prevL = eps; /* r1[0] = epsilon-filled "zeros_matrix" (elf:40-41,48) */
            for (int d = 0; d < D; ++d) {
                float v = col[(size_t)d * hw];
                float o = v < eps ? eps : (v > hi ? hi : v);        /* elf:32 */
                float x = logf(1.0f - o);                            /* elf:34 */
                float xp = logf(o);                                   /* elf:35 */
                out[(size_t)d * hw] = expf(prevL + xp);              /* elf:54-56 */
                acc += (double)x;                                     /* elf:37 */
                prevL = (float)acc;
            }
            out[(size_t)D * hw] = expf(prevL + eps);                 /* r2 last = eps (elf:51) */
        }
}

/* P6 tail: probs[:, :-1].sum(1).flip(1)  (elf:81) -> [B,H,W] */
void orc_project(const float *T, int B, int D, int H, int W, float *proj)
{
    size_t hw = (size_t)H * W;
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                double s = 0.0;
                for (int d = 0; d < D; ++d) s += (double)T[((size_t)b * (D + 1) + d) * hw + (size_t)(H - 1 - y) * W + x];
                proj[(size_t)b * hw + (size_t)y * W + x] = (float)s;
            }
}

/* P7: SupervisedLoss (models/supervised_part.py:68-72): bilinear 1/2 downsample with
 * align_corners=True of masks[B,2S,2S] -> [B,S,S]; loss = sum((proj-m)^2)/(2B).
 * ATen upsample_bilinear2d (align_corners): src = dst * (in-1)/(out-1); lambda in float. */
void orc_mask_downsample(const float *mask, int B, int Hin, int Win, int Hout, int Wout, float *m)
{
    float sh = Hout > 1 ? (float)(Hin - 1) / (float)(Hout - 1) : 0.0f;
    float sw = Wout > 1 ? (float)(Win - 1) / (float)(Wout - 1) : 0.0f;
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < Hout; ++y) {
            float fy = sh * (float)y;
            int y0 = (int)fy; int y1 = y0 + (y0 < Hin - 1 ? 1 : 0);
            float ly = fy - (float)y0, hy = 1.0f - ly;
            for (int x = 0; x < Wout; ++x) {
                float fx = sw * (float)x;
                int x0 = (int)fx; int x1 = x0 + (x0 < Win - 1 ? 1 : 0);
                float lx = fx - (float)x0, hx = 1.0f - lx;
                const float *mb = mask + (size_t)b * Hin * Win;
                float v = hy * (hx * mb[(size_t)y0 * Win + x0] + lx * mb[(size_t)y0 * Win + x1]) +
                          ly * (hx * mb[(size_t)y1 * Win + x0] + lx * mb[(size_t)y1 * Win + x1]);
                m[((size_t)b * Hout + y) * Wout + x] = v;
            }
        }
}

double orc_sup_loss(const float *proj, const float *mask, int B, int S)
{
    size_t n = (size_t)B * S * S;
    float *m = (float *)malloc(sizeof(float) * n);
    orc_mask_downsample(mask, B, 2 * S, 2 * S, S, S, m);
    double s = 0.0;
    for (size_t i = 0; i < n; ++i) { float d = proj[i] - m[i]; s += (double)(d * d); }
    free(m);
    return s / (2.0 * B);
}

/* Whole forward, literal semantics + shims S0/S1: cloud -> silhouette [B,S,S].
 * Optional outputs cam[B,N,3] (may be NULL). */
void orc_forward(const float *p, const float *q, const float *scale, int B, int N, int S,
                 const float *taps, int ntaps, int axis_mask, int fixed_weights, float *cam_out, float *proj)
{
    size_t vol = (size_t)S * S * S;
    float *cam = cam_out ? cam_out : (float *)malloc(sizeof(float) * (size_t)B * N * 3);
    float *V = (float *)malloc(sizeof(float) * vol);
    float *Vs = (float *)malloc(sizeof(float) * vol);
    float *T = (float *)malloc(sizeof(float) * (size_t)(S + 1) * S * S);
    orc_transform(p, q, B, N, cam);
    for (int b = 0; b < B; ++b) {
        orc_splat(cam + (size_t)b * N * 3, 1, N, S, fixed_weights, V);
        orc_smooth(V, 1, S, taps, ntaps, axis_mask, scale ? scale + b : NULL, Vs);
        orc_termination(Vs, 1, S, S, S, 1e-5f, T);
        orc_project(T, 1, S, S, S, proj + (size_t)b * S * S);
    }
    free(V); free(Vs); free(T);
    if (!cam_out) free(cam);
}

/* ---- analytic backward of P1..P6 (what autograd computes for the reference), double precision.
 * dproj[B,S,S] -> dp[B,N,3], dq[B,4], dscale[B] (dscale may be NULL when scale is NULL). */
static void hamilton_d(const double a[4], const double b[4], double r[4])
{
    r[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    r[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    r[2] = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
    r[3] = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
}
/* r = a (x) b ; given dr accumulate da, db */
static void hamilton_bwd(const double a[4], const double b[4], const double dr[4], double da[4], double db[4])
{
    da[0] += dr[0] * b[0] + dr[1] * b[1] + dr[2] * b[2] + dr[3] * b[3];
    da[1] += -dr[0] * b[1] + dr[1] * b[0] - dr[2] * b[3] + dr[3] * b[2];
    da[2] += -dr[0] * b[2] + dr[1] * b[3] + dr[2] * b[0] - dr[3] * b[1];
    da[3] += -dr[0] * b[3] - dr[1] * b[2] + dr[2] * b[1] + dr[3] * b[0];
    db[0] += dr[0] * a[0] + dr[1] * a[1] + dr[2] * a[2] + dr[3] * a[3];
    db[1] += -dr[0] * a[1] + dr[1] * a[0] + dr[2] * a[3] - dr[3] * a[2];
    db[2] += -dr[0] * a[2] - dr[1] * a[3] + dr[2] * a[0] + dr[3] * a[1];
    db[3] += -dr[0] * a[3] + dr[1] * a[2] - dr[2] * a[1] + dr[3] * a[0];
}

/* dcam[B,N,3] -> dp, dq  (backward of orc_transform) */
void orc_transform_bwd(const float *p, const float *q, const float *dcam, int B, int N, float *dp, float *dq)
{
    for (int b = 0; b < B; ++b) {
        double qr[4], qn[4], qs[4], dqn[4] = {0, 0, 0, 0};
        double s = 0;
        for (int i = 0; i < 4; ++i) { qr[i] = q[4 * b + i]; s += qr[i] * qr[i]; }
        double nrm = sqrt(s); double den = nrm < 1e-12 ? 1e-12 : nrm;
        for (int i = 0; i < 4; ++i) qn[i] = qr[i] / den;
        qs[0] = qn[0]; qs[1] = -qn[1]; qs[2] = -qn[2]; qs[3] = -qn[3];
        for (int n = 0; n < N; ++n) {
            size_t o = ((size_t)b * N + n) * 3;
            double p4[4] = {0, p[o], p[o + 1], p[o + 2]}, t[4], r[4];
            hamilton_d(qn, p4, t); hamilton_d(t, qs, r);
            double z = r[1], y = r[2], x = r[3], dn = z + 2.0;
            double dz = dcam[o], dyo = dcam[o + 1], dxo = dcam[o + 2];
            /* xo = x*f/dn ; yo = y*f/dn ; z passthrough */
            double dr[4] = {0, 0, 0, 0};
            dr[3] = dxo * 1.875 / dn;
            dr[2] = dyo * 1.875 / dn;
            dr[1] = dz - dxo * x * 1.875 / (dn * dn) - dyo * y * 1.875 / (dn * dn);
            double dt[4] = {0, 0, 0, 0}, dqs[4] = {0, 0, 0, 0}, dp4[4] = {0, 0, 0, 0};
            hamilton_bwd(t, qs, dr, dt, dqs);
            hamilton_bwd(qn, p4, dt, dqn, dp4);
            dqn[0] += dqs[0]; dqn[1] -= dqs[1]; dqn[2] -= dqs[2]; dqn[3] -= dqs[3];
            dp[o] = (float)dp4[1]; dp[o + 1] = (float)dp4[2]; dp[o + 2] = (float)dp4[3];
        }
        /* normalize backward: qn = qr/den */
        double dot = 0; for (int i = 0; i < 4; ++i) dot += dqn[i] * qr[i];
        for (int i = 0; i < 4; ++i) {
            double g = dqn[i] / den;
            if (nrm >= 1e-12) g -= dot * qr[i] / (den * den * nrm);
            dq[4 * b + i] = (float)g;
        }
    }
}

/* dproj -> dcam[B,N,3], dscale[B]; depth-only or multi-axis smoothing (chained order x,y,depth). */
void orc_project_bwd(const float *cam, const float *scale, const float *dproj, int B, int N, int S,
                     const float *taps, int ntaps, int axis_mask, int fixed_weights, float *dcam, float *dscale)
{
    size_t vol = (size_t)S * S * S, hw = (size_t)S * S;
    const float sm1 = (float)S - 1.0f;
    const float eps = 1e-5f, hi = (float)(1.0 - 1e-5);
    float *Vraw = (float *)malloc(sizeof(float) * vol);
    float *Vc = (float *)malloc(sizeof(float) * vol);
    float *Sm = (float *)malloc(sizeof(float) * vol);
    double *G = (double *)malloc(sizeof(double) * vol);
    double *G2 = (double *)malloc(sizeof(double) * vol);
    int half = ntaps / 2;
    for (int b = 0; b < B; ++b) {
        splat_raw(cam, b, N, S, fixed_weights, Vraw);
        for (size_t i = 0; i < vol; ++i) { float v = Vraw[i]; Vc[i] = v < 0 ? 0 : (v > 1 ? 1 : v); }
        orc_smooth(Vc, 1, S, taps, ntaps, axis_mask, NULL, Sm);
        double ds = 0.0;
        for (size_t r = 0; r < hw; ++r) {
            int y = (int)(r / S), x = (int)(r % S);
            double g = dproj[(size_t)b * hw + (size_t)(S - 1 - y) * S + x]; /* flip(1) */
            /* forward per ray in double on the float-rounded operands */
            double Lprev = (double)eps, acc = 0.0;
            double *Tn = (double *)G2; /* reuse as scratch: T[d] for this ray at G2[d] (first S entries) */
            float oarr[4096]; double suffix = 0.0;
            for (int d = 0; d < S; ++d) {
                float sv = Sm[(size_t)d * hw + r];
                float c = sv;
                if (scale) { c = sv * scale[b]; c = c < 0 ? 0 : (c > 1 ? 1 : c); }
                float o = c < eps ? eps : (c > hi ? hi : c);
                oarr[d] = o;
                Tn[d] = exp((double)(float)Lprev + (double)logf(o));
                acc += (double)logf(1.0f - o);
                Lprev = acc;
            }
            for (int d = S - 1; d >= 0; --d) {
                /* do = g*T[d]/o - g*sum_{m>d}T[m]/(1-o) */
                float o = oarr[d];
                double d_o = g * Tn[d] / (double)o - g * suffix / (double)(1.0f - o);
                suffix += Tn[d];
                float sv = Sm[(size_t)d * hw + r];
                float c = sv;
                double dsv = d_o;
                if (scale) {
                    float cs = sv * scale[b];
                    float cc = cs < 0 ? 0 : (cs > 1 ? 1 : cs);
                    c = cc;
                    if (!(c >= eps && c <= hi)) dsv = 0.0;
                    if (!(cs >= 0.0f && cs <= 1.0f)) dsv = 0.0;
                    ds += dsv * (double)sv;
                    dsv *= (double)scale[b];
                } else {
                    if (!(c >= eps && c <= hi)) dsv = 0.0;
                }
                G[(size_t)d * hw + r] = dsv;
            }
        }
        if (dscale) dscale[b] = (float)ds;
        /* conv transpose, reverse chained order: depth, y, x */
        int order[3] = {0, 1, 2};
        for (int oi = 0; oi < 3; ++oi) {
            int axis = order[oi];
            if (!(axis_mask & (1 << axis))) continue;
            size_t stride = axis == 0 ? hw : (axis == 1 ? (size_t)S : 1);
            for (int z = 0; z < S; ++z) for (int y = 0; y < S; ++y) for (int x = 0; x < S; ++x) {
                int pos = axis == 0 ? z : (axis == 1 ? y : x);
                size_t base = ((size_t)z * S + y) * S + x;
                double a = 0.0;
                /* out[p] = sum_t taps[t]*in[p+t-half]  =>  din[q] = sum_t taps[t]*dout[q-t+half] */
                for (int t = 0; t < ntaps; ++t) {
                    int pp = pos - t + half;
                    if (pp < 0 || pp >= S) continue;
                    a += (double)taps[t] * G[base + (ptrdiff_t)(pp - pos) * (ptrdiff_t)stride];
                }
                G2[base] = a;
            }
            memcpy(G, G2, sizeof(double) * vol);
        }
        for (size_t i = 0; i < vol; ++i) if (!(Vraw[i] >= 0.0f && Vraw[i] <= 1.0f)) G[i] = 0.0;
        for (int n = 0; n < N; ++n) {
            size_t o = ((size_t)b * N + n) * 3;
            const float *c = cam + o;
            dcam[o] = dcam[o + 1] = dcam[o + 2] = 0.0f;
            if (!in_bounds(c)) continue;
            float g[3], f[3]; double w[3][2], dw[3][2];
            for (int a = 0; a < 3; ++a) {
                g[a] = sm1 * (c[a] + 0.5f); f[a] = floorf(g[a]);
                w[a][1] = (double)(g[a] - f[a]); dw[a][1] = 1.0;
                w[a][0] = fixed_weights ? (double)(1.0f - (g[a] - f[a])) : (double)((1.0f - g[a]) - f[a]);
                dw[a][0] = -1.0;
            }
            double dg[3] = {0, 0, 0};
            for (int c8 = 0; c8 < 8; ++c8) {
                int ci = (c8 >> 2) & 1, cj = (c8 >> 1) & 1, ck = c8 & 1;
                size_t idx = ((size_t)((int)f[0] + ci) * S + ((int)f[1] + cj)) * S + ((int)f[2] + ck);
                double gv = G[idx];
                dg[0] += gv * dw[0][ci] * w[1][cj] * w[2][ck];
                dg[1] += gv * w[0][ci] * dw[1][cj] * w[2][ck];
                dg[2] += gv * w[0][ci] * w[1][cj] * dw[2][ck];
            }
            for (int a = 0; a < 3; ++a) dcam[o + a] = (float)(dg[a] * (double)sm1);
        }
    }
    free(Vraw); free(Vc); free(Sm); free(G); free(G2);
}

/* d(sup loss)/dproj = (proj - m)/B * gout */
void orc_sup_loss_bwd(const float *proj, const float *mask, int B, int S, float gout, float *dproj)
{
    size_t n = (size_t)B * S * S;
    float *m = (float *)malloc(sizeof(float) * n);
    orc_mask_downsample(mask, B, 2 * S, 2 * S, S, S, m);
    for (size_t i = 0; i < n; ++i) dproj[i] = (float)((double)(proj[i] - m[i]) / (double)B * (double)gout);
    free(m);
}

/* Chamfer NN (new capability, no reference implementation -- "parity unpinned"):
 * d1[i] = min_j ||a_i - b_j||^2, i1[i] = argmin (lowest j on ties); direction A->B only.
 * The squared distance is DEFINED as the fused chain fma(dz, dz, fma(dy, dy, dx * dx)) (round 4; one rounding less than the
 * unfused sum of round 1-3): both sides evaluate exactly this, so distances and the tie rule stay bit-exact. */
#if defined(__x86_64__)
__attribute__((target("fma")))
#endif
void orc_chamfer_nn(const float *a, const float *bb, int B, int N, int M, float *d1, int32_t *i1)
{
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < N; ++i) {
            const float *pa = a + ((size_t)b * N + i) * 3;
            float best = INFINITY; int bi = -1;
            for (int j = 0; j < M; ++j) {
                const float *pb = bb + ((size_t)b * M + j) * 3;
                float dx = pa[0] - pb[0], dy = pa[1] - pb[1], dz = pa[2] - pb[2];
                float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                if (d < best) { best = d; bi = j; }
            }
            d1[(size_t)b * N + i] = best; i1[(size_t)b * N + i] = bi;
        }
}
