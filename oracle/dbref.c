/*
 * ORACLE — test infrastructure, not product code.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; nothing under deepbinner_amd/ does.
 *
 * Plain-C fp32 restatement of the reference's hot path:
 *   - the inference graph of deepbinner/network_architecture.py:18-95 as model.predict evaluates
 *     it (deepbinner/classify.py:361) — TensorFlow semantics as documented in
 *     oracle/network_ref.py (SAME padding puts the odd pad element on the right; average pooling
 *     divides by the number of valid taps; BN after ReLU/MaxPool);
 *   - normalise (deepbinner/trim_signal.py:61-69) and the window/merge/renormalise/call logic of
 *     call_batch (deepbinner/classify.py:325-393, 285-295).
 * It is the "port" CPU baseline timed by bench.py (threads = OpenMP threads used) and is itself
 * checked against the NumPy oracle and the reference's golden answers in tests/.
 *
 * Weights: the canonical flat blob of deepbinner_amd/model_format.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NCONV 20
#define NBN 7
#define WINDOW 1024

typedef struct { int k, cin, cout, stride, same; } conv_spec;

static const conv_spec SPECS[NCONV] = {
    {3, 1, 48, 2, 1},  {3, 48, 48, 1, 1}, {3, 48, 48, 1, 1}, {3, 48, 48, 1, 1},
    {1, 48, 16, 1, 0}, {3, 16, 48, 1, 1}, {3, 48, 48, 1, 1}, {3, 48, 48, 1, 1},
    {3, 48, 48, 1, 1}, {1, 48, 48, 1, 1}, {1, 48, 48, 1, 1}, {1, 48, 16, 1, 1},
    {3, 16, 48, 1, 1}, {1, 48, 16, 1, 1}, {3, 16, 48, 1, 1}, {3, 48, 48, 1, 1},
    {3, 192, 48, 2, 1}, {3, 48, 48, 1, 1}, {3, 48, 48, 1, 1}, {1, 48, -1, 1, 0}};
static const int BN_C[NBN] = {48, 48, 48, 48, 192, 48, 48};

typedef struct {
    int n_classes;
    const float* kernel[NCONV];
    const float* bias[NCONV];
    int cout[NCONV];
    float* bn_scale[NBN];
    float* bn_shift[NBN];
    float* owned;
} dbref_model;

int64_t dbref_param_count(int n_classes) {
    int64_t n = 0;
    for (int i = 0; i < NCONV; ++i) {
        int cout = SPECS[i].cout < 0 ? n_classes : SPECS[i].cout;
        n += (int64_t)SPECS[i].k * SPECS[i].cin * cout + cout;
    }
    for (int i = 0; i < NBN; ++i) n += 4 * BN_C[i];
    return n;
}

dbref_model* dbref_create(const float* blob, int64_t n_floats, int n_classes) {
    if (n_floats != dbref_param_count(n_classes)) return NULL;
    dbref_model* m = (dbref_model*)calloc(1, sizeof(dbref_model));
    m->owned = (float*)malloc((size_t)n_floats * sizeof(float));
    memcpy(m->owned, blob, (size_t)n_floats * sizeof(float));
    m->n_classes = n_classes;
    const float* p = m->owned;
    for (int i = 0; i < NCONV; ++i) {
        int cout = SPECS[i].cout < 0 ? n_classes : SPECS[i].cout;
        m->cout[i] = cout;
        m->kernel[i] = p;
        p += (size_t)SPECS[i].k * SPECS[i].cin * cout;
        m->bias[i] = p;
        p += cout;
    }
    for (int i = 0; i < NBN; ++i) {
        int c = BN_C[i];
        m->bn_scale[i] = (float*)malloc(sizeof(float) * c);
        m->bn_shift[i] = (float*)malloc(sizeof(float) * c);
        for (int j = 0; j < c; ++j) {
            float scale = p[j] / sqrtf(p[3 * c + j] + 1e-3f);
            m->bn_scale[i][j] = scale;
            m->bn_shift[i][j] = p[c + j] - p[2 * c + j] * scale;
        }
        p += 4 * c;
    }
    return m;
}

void dbref_destroy(dbref_model* m) {
    if (!m) return;
    for (int i = 0; i < NBN; ++i) {
        free(m->bn_scale[i]);
        free(m->bn_shift[i]);
    }
    free(m->owned);
    free(m);
}

/* y[Lout][cout] = relu(conv(x[L][cin])) ; returns Lout */
static int conv_relu(const dbref_model* m, int idx, const float* x, int L, float* y) {
    const conv_spec s = SPECS[idx];
    const int cout = m->cout[idx];
    int Lout, left = 0;
    if (s.same) {
        Lout = (L + s.stride - 1) / s.stride;
        int total = (Lout - 1) * s.stride + s.k - L;
        if (total < 0) total = 0;
        left = total / 2;
    } else {
        Lout = (L - s.k) / s.stride + 1;
    }
    for (int p = 0; p < Lout; ++p) {
        float* out = y + (size_t)p * cout;
        for (int c = 0; c < cout; ++c) out[c] = m->bias[idx][c];
        for (int j = 0; j < s.k; ++j) {
            int q = p * s.stride + j - left;
            if (q < 0 || q >= L) continue;
            const float* xin = x + (size_t)q * s.cin;
            const float* w = m->kernel[idx] + (size_t)j * s.cin * cout;
            for (int ci = 0; ci < s.cin; ++ci) {
                const float xv = xin[ci];
                const float* wr = w + (size_t)ci * cout;
                for (int c = 0; c < cout; ++c) out[c] += xv * wr[c];
            }
        }
        for (int c = 0; c < cout; ++c) out[c] = out[c] > 0.f ? out[c] : 0.f;
    }
    return Lout;
}

static int maxpool2(float* x, int L, int C) {
    int half = L / 2;
    for (int p = 0; p < half; ++p)
        for (int c = 0; c < C; ++c) {
            float a = x[(size_t)(2 * p) * C + c], b = x[(size_t)(2 * p + 1) * C + c];
            x[(size_t)p * C + c] = a > b ? a : b;
        }
    return half;
}

static void batchnorm(const dbref_model* m, int idx, float* x, int L, int C) {
    for (int p = 0; p < L; ++p)
        for (int c = 0; c < C; ++c)
            x[(size_t)p * C + c] = x[(size_t)p * C + c] * m->bn_scale[idx][c] + m->bn_shift[idx][c];
}

/* one window: x[1024] -> probs[n_classes]; scratch: 4 buffers of 512*48 floats + 1 of 64*192 */
static void forward_one(const dbref_model* m, const float* x, float* probs, float* s0, float* s1,
                        float* s2, float* cat) {
    int L = conv_relu(m, 0, x, WINDOW, s0);                 /* network_architecture.py:28 */
    batchnorm(m, 0, s0, L, 48);
    L = conv_relu(m, 1, s0, L, s1);                         /* :34-36 */
    L = conv_relu(m, 2, s1, L, s0);
    L = conv_relu(m, 3, s0, L, s1);
    L = maxpool2(s1, L, 48);                                /* :37 */
    batchnorm(m, 1, s1, L, 48);
    L = conv_relu(m, 4, s1, L, s0);                         /* :43 */
    L = conv_relu(m, 5, s0, L, s1);                         /* :46-47 */
    L = conv_relu(m, 6, s1, L, s0);
    L = maxpool2(s0, L, 48);
    batchnorm(m, 2, s0, L, 48);
    L = conv_relu(m, 7, s0, L, s1);                         /* :54-55 */
    L = conv_relu(m, 8, s1, L, s0);
    L = maxpool2(s0, L, 48);
    batchnorm(m, 3, s0, L, 48);                             /* s0: 64 x 48 */
    /* inception (:62-70) */
    for (int p = 0; p < L; ++p)
        for (int c = 0; c < 48; ++c) {
            float sum = s0[(size_t)p * 48 + c];
            int cnt = 1;
            if (p > 0) { sum += s0[(size_t)(p - 1) * 48 + c]; ++cnt; }
            if (p < L - 1) { sum += s0[(size_t)(p + 1) * 48 + c]; ++cnt; }
            s1[(size_t)p * 48 + c] = sum / (float)cnt;
        }
    conv_relu(m, 9, s1, L, s2);
    for (int p = 0; p < L; ++p) memcpy(cat + (size_t)p * 192, s2 + (size_t)p * 48, 48 * sizeof(float));
    conv_relu(m, 10, s0, L, s2);
    for (int p = 0; p < L; ++p) memcpy(cat + (size_t)p * 192 + 48, s2 + (size_t)p * 48, 48 * sizeof(float));
    conv_relu(m, 11, s0, L, s1);
    conv_relu(m, 12, s1, L, s2);
    for (int p = 0; p < L; ++p) memcpy(cat + (size_t)p * 192 + 96, s2 + (size_t)p * 48, 48 * sizeof(float));
    conv_relu(m, 13, s0, L, s1);
    conv_relu(m, 14, s1, L, s2);
    conv_relu(m, 15, s2, L, s1);
    for (int p = 0; p < L; ++p) memcpy(cat + (size_t)p * 192 + 144, s1 + (size_t)p * 48, 48 * sizeof(float));
    L = maxpool2(cat, L, 192);                              /* :71 */
    batchnorm(m, 4, cat, L, 192);
    L = conv_relu(m, 16, cat, L, s0);                       /* :77 */
    batchnorm(m, 5, s0, L, 48);
    L = conv_relu(m, 17, s0, L, s1);                        /* :83-84 */
    L = conv_relu(m, 18, s1, L, s0);
    L = maxpool2(s0, L, 48);
    batchnorm(m, 6, s0, L, 48);
    L = conv_relu(m, 19, s0, L, s1);                        /* :91 */
    const int C = m->n_classes;
    float logits[64];
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) {
        float sum = 0.f;
        for (int p = 0; p < L; ++p) sum += s1[(size_t)p * C + c];
        logits[c] = sum / (float)L;                         /* :92 */
        if (logits[c] > mx) mx = logits[c];
    }
    float den = 0.f;
    for (int c = 0; c < C; ++c) {
        logits[c] = expf(logits[c] - mx);
        den += logits[c];
    }
    for (int c = 0; c < C; ++c) probs[c] = logits[c] / den;  /* :93 */
}

/* x[n][1024] -> probs[n][n_classes]; threads <= 0 means "all OpenMP threads". Returns threads used. */
int dbref_predict(const dbref_model* m, const float* x, int64_t n, float* probs, int threads) {
    int used = 1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    used = omp_get_max_threads();
#pragma omp parallel
#endif
    {
        float* s0 = (float*)malloc(sizeof(float) * 512 * 48);
        float* s1 = (float*)malloc(sizeof(float) * 512 * 48);
        float* s2 = (float*)malloc(sizeof(float) * 512 * 48);
        float* cat = (float*)malloc(sizeof(float) * 64 * 192);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
        for (int64_t i = 0; i < n; ++i)
            forward_one(m, x + i * WINDOW, probs + i * m->n_classes, s0, s1, s2, cat);
        free(s0); free(s1); free(s2); free(cat);
    }
    return used;
}

/* trim_signal.py:61-69 + classify.py:337-358: windows[read*steps+step][1024] */
void dbref_windows(const int16_t* samples, const int64_t* offsets, int64_t n_reads, int side,
                   int scan_size, float* windows) {
    const int steps = scan_size / (WINDOW / 2);
    for (int64_t r = 0; r < n_reads; ++r) {
        const int64_t len = offsets[r + 1] - offsets[r];
        const int16_t* sig = samples + offsets[r];
        for (int s = 0; s < steps; ++s) {
            const int64_t st = (int64_t)s * (WINDOW / 2), en = st + WINDOW;
            int64_t a, b;
            if (side == 0) { a = st < len ? st : len; b = en < len ? en : len; }
            else { a = len - en > 0 ? len - en : 0; b = len - st > 0 ? len - st : 0; }
            const int cnt = (int)(b - a);
            float* out = windows + ((size_t)r * steps + s) * WINDOW;
            memset(out, 0, sizeof(float) * WINDOW);
            if (cnt == 0) continue;
            double sum = 0.0;
            for (int k = 0; k < cnt; ++k) sum += sig[a + k];
            const double mean = sum / cnt;
            double var = 0.0;
            for (int k = 0; k < cnt; ++k) { double d = sig[a + k] - mean; var += d * d; }
            const double sd = sqrt(var / cnt);
            const int pad = side == 0 ? 0 : WINDOW - cnt;
            for (int k = 0; k < cnt; ++k) {
                double d = sig[a + k] - mean;
                out[pad + k] = (float)(sd > 0.0 ? d / sd : d);
            }
        }
    }
}

/* classify.py:363-393 + 285-295 */
void dbref_merge(const float* wprobs, int64_t n_reads, int steps, int C, double score_diff,
                 float* probs, int32_t* calls) {
    for (int64_t r = 0; r < n_reads; ++r) {
        double p[64];
        const float* src = wprobs + (size_t)r * steps * C;
        float merged[64];
        for (int c = 0; c < C; ++c) merged[c] = src[c];
        for (int s = 1; s < steps; ++s)
            for (int c = 0; c < C; ++c) {
                float v = src[(size_t)s * C + c];
                if (c == 0) { if (v < merged[0]) merged[0] = v; }
                else if (v > merged[c]) merged[c] = v;
            }
        double rest = 0.0;
        for (int c = 1; c < C; ++c) rest += (double)merged[c];
        const double factor = (1.0 - (double)merged[0]) / rest;
        p[0] = merged[0];
        for (int c = 1; c < C; ++c) p[c] = (double)merged[c] * factor;
        int best = 0;
        for (int c = 1; c < C; ++c) if (p[c] > p[best]) best = c;
        double second = -1.0;
        for (int c = 0; c < C; ++c) if (c != best && p[c] > second) second = p[c];
        calls[r] = (best != 0 && p[best] - second >= score_diff) ? best : 0;
        for (int c = 0; c < C; ++c) probs[(size_t)r * C + c] = (float)p[c];
    }
}
