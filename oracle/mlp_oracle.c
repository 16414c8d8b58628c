/*
 * mlp_oracle.c -- TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
 *
 * Serial fp64 CPU restatement of the reference's mini-batch SGD hot path
 * (asheptunov/graph-neural-net, Java).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the HIP product path never does.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference holds no tests, golden vectors or
 * fixtures for this path and cannot be executed here (no JDK in
 * the image).  What pins this restatement instead (tests/test_oracle.py):
 *   - java.util.Random known-answer values from the documented LCG,
 *   - an independent numpy matrix-form oracle that must agree to 1e-12 relative,
 *   - a finite-difference gradient check,
 *   - hand-worked 2-2-2 vectors.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/src).  SCE = SoftmaxCrossEntropyNeuralNet.java, GNN = GeneralNeuralNet.java,
 * MT = MNISTTrainer.java, NNT = NeuralNetTrainer.java.
 *
 * Arithmetic is IEEE binary64, sequential left-to-right sums, no FMA contraction
 * (compile with -ffp-contract=off): Java never fuses a multiply and an add.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * java.util.Random (JDK; used at SCE:111,149,335  GNN:130,168,355  NNT:42,152).
 * 48-bit LCG as specified by the class's Javadoc.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t seed;            /* 48 bits */
    double next_next_gaussian;
    int have_next_next_gaussian;
} jrandom;

#define JR_MULT 0x5DEECE66DULL
#define JR_ADD 0xBULL
#define JR_MASK ((1ULL << 48) - 1)

void jrandom_seed(jrandom *r, int64_t seed) {
    r->seed = ((uint64_t)seed ^ JR_MULT) & JR_MASK;
    r->have_next_next_gaussian = 0;
    r->next_next_gaussian = 0.0;
}

static int32_t jr_next(jrandom *r, int bits) {
    r->seed = (r->seed * JR_MULT + JR_ADD) & JR_MASK;
    /* (int)(seed >>> (48 - bits)): truncation to the low 32 bits, two's complement */
    return (int32_t)(uint32_t)(r->seed >> (48 - bits));
}

int32_t jrandom_next_int(jrandom *r) { return jr_next(r, 32); }

int32_t jrandom_next_int_bound(jrandom *r, int32_t bound) {
    if (bound <= 0) return -1; /* Java throws IllegalArgumentException */
    int32_t m = bound - 1;
    int32_t rr = jr_next(r, 31);
    if ((bound & m) == 0) { /* power of two */
        return (int32_t)(((int64_t)bound * (int64_t)rr) >> 31);
    }
    /* for (int u = r; u - (r = u % bound) + m < 0; u = next(31));  with int32 wrap-around */
    int32_t u = rr;
    for (;;) {
        rr = u % bound;
        int32_t t = (int32_t)((uint32_t)u - (uint32_t)rr + (uint32_t)m);
        if (t >= 0) break;
        u = jr_next(r, 31);
    }
    return rr;
}

double jrandom_next_double(jrandom *r) {
    int64_t hi = (int64_t)jr_next(r, 26);
    int64_t lo = (int64_t)jr_next(r, 27);
    return (double)((hi << 27) + lo) * 0x1.0p-53;
}

/* Marsaglia polar method, cached second value (StrictMath.log / StrictMath.sqrt in the JDK). */
double jrandom_next_gaussian(jrandom *r) {
    if (r->have_next_next_gaussian) {
        r->have_next_next_gaussian = 0;
        return r->next_next_gaussian;
    }
    double v1, v2, s;
    do {
        v1 = 2 * jrandom_next_double(r) - 1;
        v2 = 2 * jrandom_next_double(r) - 1;
        s = v1 * v1 + v2 * v2;
    } while (s >= 1 || s == 0);
    double multiplier = sqrt(-2 * log(s) / s);
    r->next_next_gaussian = v2 * multiplier;
    r->have_next_next_gaussian = 1;
    return v1 * multiplier;
}

/* ------------------------------------------------------------------------------------------
 * L0 plug functions as a closed enum (the reference takes arbitrary lambdas:
 * ActivationFunction.java:14, ActivationPrime.java:14, LossFunction.java:16,
 * LossFunctionPrime.java:13).  Codes are shared with include/gnn_mlp.h.
 * ---------------------------------------------------------------------------------------- */
enum { ACT_LEAKY_RELU = 0, ACT_SIGMOID = 1, ACT_TANH = 2, ACT_RELU = 3, ACT_IDENTITY = 4 };
enum { OUT_SOFTMAX_CE = 0, OUT_ACT_LOSS = 1 };
enum { LOSS_HALF_SQUARED = 0 };

static double act_f(int kind, double a) {
    switch (kind) {
    case ACT_LEAKY_RELU: return (a > 0) ? a : 0.01 * a;      /* MT:234 */
    case ACT_SIGMOID: return 1.0 / (1.0 + exp(-a));          /* doc/backprop.pdf section 1 */
    case ACT_TANH: return tanh(a);
    case ACT_RELU: return (a > 0) ? a : 0.0;
    default: return a;
    }
}

/* derivative as a function of the PRE-activation sum (SCE:277, GNN:270) */
static double act_prime(int kind, double a) {
    switch (kind) {
    case ACT_LEAKY_RELU: return (a <= 0.0) ? 0.01 : 1.0;     /* MT:235 -- note a==0 -> 0.01 */
    case ACT_SIGMOID: { double s = 1.0 / (1.0 + exp(-a)); return s * (1.0 - s); }
    case ACT_TANH: { double t = tanh(a); return 1.0 - t * t; }
    case ACT_RELU: return (a <= 0.0) ? 0.0 : 1.0;
    default: return 1.0;
    }
}

static double loss_f(int kind, double calculated, double expected) {
    (void)kind; /* half squared error: doc/backprop.pdf section 2 */
    double d = calculated - expected;
    return 0.5 * d * d;
}

static double loss_prime(int kind, double calculated, double expected) {
    (void)kind;
    return calculated - expected;
}

/* ------------------------------------------------------------------------------------------
 * Net state (SCE:13-24, GNN:13-28)
 * ---------------------------------------------------------------------------------------- */
typedef struct oracle_net {
    int n_layers;       /* layerDims.length */
    int *dims;          /* layerDims */
    double **neurons;   /* neurons[l]: PRE-activation sums; neurons[0] aliases the caller's input */
    double *input_copy; /* stands in for the aliased input array (SCE:167-168) */
    double **weights;   /* weights[l][i*d_{l+1}+k], row-major [in][out] (SCE:44-47) */
    double **prev;      /* previousUpdate[l] */
    int time;           /* SCE:23 */
    jrandom random;     /* SCE:18 */
    int out_kind, inner_act, last_act, loss;
    long n_params;
    int alloc_per_sample; /* 1: calloc fresh per-sample gradient arrays like the Java code */
} oracle_net;

/* ctor + appendLayer: SCE:103-128, SCE:139-156 (GNN:112-147, GNN:158-175) */
oracle_net *oracle_create(const int32_t *dims, int n_layers, int out_kind, int inner_act,
                          int last_act, int loss, int64_t seed) {
    if (!dims || n_layers < 2) return NULL;
    for (int i = 0; i < n_layers; i++) if (dims[i] <= 0) return NULL;
    oracle_net *n = (oracle_net *)calloc(1, sizeof(*n));
    n->n_layers = n_layers;
    n->dims = (int *)malloc(sizeof(int) * n_layers);
    memcpy(n->dims, dims, sizeof(int) * n_layers);
    n->neurons = (double **)calloc(n_layers, sizeof(double *));
    n->weights = (double **)calloc(n_layers - 1, sizeof(double *));
    n->prev = (double **)calloc(n_layers - 1, sizeof(double *));
    n->out_kind = out_kind; n->inner_act = inner_act; n->last_act = last_act; n->loss = loss;
    n->time = 0;
    n->alloc_per_sample = 1;
    jrandom_seed(&n->random, seed); /* new Random(1) at SCE:111 */
    n->input_copy = (double *)calloc(dims[0], sizeof(double));
    for (int l = 0; l < n_layers; l++) {
        int dim = dims[l];
        if (l > 0) {
            int in = dims[l - 1];
            double *w = (double *)malloc(sizeof(double) * in * dim);
            for (int i = 0; i < in; i++)
                for (int j = 0; j < dim; j++)
                    w[(size_t)i * dim + j] = jrandom_next_double(&n->random) - 0.5; /* SCE:149 */
            n->weights[l - 1] = w;
            n->prev[l - 1] = (double *)calloc((size_t)in * dim, sizeof(double)); /* SCE:153 */
            n->n_params += (long)in * dim;
        }
        n->neurons[l] = (l == 0) ? n->input_copy : (double *)calloc(dim, sizeof(double));
    }
    return n;
}

void oracle_destroy(oracle_net *n) {
    if (!n) return;
    for (int l = 1; l < n->n_layers; l++) free(n->neurons[l]);
    for (int l = 0; l < n->n_layers - 1; l++) { free(n->weights[l]); free(n->prev[l]); }
    free(n->input_copy); free(n->neurons); free(n->weights); free(n->prev); free(n->dims);
    free(n);
}

int oracle_input_dim(const oracle_net *n) { return n->dims[0]; }                 /* SCE:383 */
int oracle_output_dim(const oracle_net *n) { return n->dims[n->n_layers - 1]; } /* SCE:392 */
long oracle_num_params(const oracle_net *n) { return n->n_params; }
int oracle_time(const oracle_net *n) { return n->time; }
void oracle_set_alloc_per_sample(oracle_net *n, int v) { n->alloc_per_sample = v; }

/* softmaxActivate: SCE:357-376.  The max IS computed and never used (SCE:361-364 vs SCE:368). */
static void softmax_activate(const double *input, int l, double *output) {
    double max = input[0];
    for (int i = 1; i < l; i++) if (input[i] > max) max = input[i];
    (void)max;
    double sum = 0;
    for (int i = 0; i < l; i++) {
        double term = exp(input[i]);
        output[i] = term;
        sum += term;
    }
    for (int i = 0; i < l; i++) output[i] /= sum;
}

/* propagate: SCE:164-198 / GNN:183-221.  out has d_{L-1} entries.
 * Quirk kept: the inner activation is applied to neurons[0], i.e. to the raw input (SCE:183-186). */
void oracle_propagate(oracle_net *n, const double *input, double *out) {
    int L = n->n_layers;
    memcpy(n->neurons[0], input, sizeof(double) * n->dims[0]); /* SCE:167-168 (aliasing) */
    for (int l = 1; l < L; l++) {
        const double *prevn = n->neurons[l - 1];
        double *cur = n->neurons[l];
        const double *w = n->weights[l - 1];
        int pd = n->dims[l - 1], cd = n->dims[l];
        for (int i = 0; i < cd; i++) cur[i] = 0.0; /* SCE:180-182 */
        double *pa = (double *)malloc(sizeof(double) * pd); /* SCE:183 */
        for (int i = 0; i < pd; i++) pa[i] = act_f(n->inner_act, prevn[i]); /* SCE:184-186 */
        for (int i = 0; i < pd; i++) {                                       /* SCE:187-192 */
            const double a = pa[i];
            const double *wr = w + (size_t)i * cd;
            for (int j = 0; j < cd; j++) cur[j] += a * wr[j];
        }
        free(pa);
    }
    const double *on = n->neurons[L - 1];
    int od = n->dims[L - 1];
    if (n->out_kind == OUT_SOFTMAX_CE) {
        softmax_activate(on, od, out); /* SCE:197 */
    } else {
        for (int i = 0; i < od; i++) out[i] = act_f(n->last_act, on[i]); /* GNN:215-218 */
    }
}

/* raw output pre-activations z_{L-1} of the last propagate (test helper; `neurons` is private) */
void oracle_last_logits(const oracle_net *n, double *out) {
    memcpy(out, n->neurons[n->n_layers - 1], sizeof(double) * n->dims[n->n_layers - 1]);
}

/* calculateLoss: SCE:207-220 / GNN:230-242 */
double oracle_loss(oracle_net *n, const double *input, const double *expected) {
    int od = n->dims[n->n_layers - 1];
    double *act = (double *)malloc(sizeof(double) * od);
    oracle_propagate(n, input, act); /* SCE:212 */
    double loss = 0;
    if (n->out_kind == OUT_SOFTMAX_CE) {
        for (int i = 0; i < od; i++) loss -= expected[i] * log(act[i]); /* SCE:216 */
    } else {
        for (int i = 0; i < od; i++) loss += loss_f(n->loss, act[i], expected[i]); /* GNN:238 */
    }
    free(act);
    return loss;
}

/* calculateWeightGradient: SCE:229-287 / GNN:251-307.
 * dEdw[l] must point to d_l*d_{l+1} doubles (the per-sample dense outer products). */
static void weight_gradient(oracle_net *n, const double *input, const double *expected,
                            double **dEdw) {
    int L = n->n_layers;
    int od = n->dims[L - 1];
    double *outact = (double *)malloc(sizeof(double) * od);
    oracle_propagate(n, input, outact); /* SCE:237 */
    double **dEdNet = (double **)calloc(L, sizeof(double *));
    double *last = (double *)calloc(od, sizeof(double));
    dEdNet[L - 1] = last;

    /* OUTPUT LAYER CASE */
    const double *jN = n->neurons[L - 1];
    const double *iN = n->neurons[L - 2];
    if (n->out_kind == OUT_SOFTMAX_CE) {
        for (int j = 0; j < od; j++) last[j] = outact[j] - expected[j]; /* SCE:249-251 */
    } else {
        for (int j = 0; j < od; j++) { /* GNN:267-271: prime takes the pre-activation z */
            double ja = act_f(n->last_act, jN[j]);
            last[j] = loss_prime(n->loss, ja, expected[j]) * act_prime(n->last_act, jN[j]);
        }
    }
    {
        int id = n->dims[L - 2];
        double *g = dEdw[L - 2];
        for (int i = 0; i < id; i++) { /* SCE:253-258 */
            double ia = act_f(n->inner_act, iN[i]);
            for (int j = 0; j < od; j++) g[(size_t)i * od + j] = last[j] * ia;
        }
    }
    /* INNER LAYER CASE: SCE:262-284 */
    for (int jL = L - 2; jL >= 1; jL--) {
        int iL = jL - 1;
        int jd = n->dims[jL], id = n->dims[iL], kd = n->dims[jL + 1];
        double *dj = (double *)calloc(jd, sizeof(double));
        dEdNet[jL] = dj;
        jN = n->neurons[jL];
        iN = n->neurons[iL];
        const double *wjk = n->weights[jL];
        const double *dk = dEdNet[jL + 1];
        for (int j = 0; j < jd; j++) { /* SCE:272-278 */
            double sum = 0;
            for (int k = 0; k < kd; k++) sum += wjk[(size_t)j * kd + k] * dk[k];
            dj[j] = sum * act_prime(n->inner_act, jN[j]);
        }
        double *g = dEdw[iL];
        for (int i = 0; i < id; i++) { /* SCE:279-283: f(i) evaluated inside the j loop */
            for (int j = 0; j < jd; j++)
                g[(size_t)i * jd + j] = dj[j] * act_f(n->inner_act, iN[i]);
        }
    }
    for (int l = 1; l < L; l++) free(dEdNet[l]);
    free(dEdNet);
    free(outact);
}

/* NeuralNet.calculateWeightGradient (NN:39) with the Map flattened layer-major, row-major. */
void oracle_weight_gradient(oracle_net *n, const double *input, const double *expected,
                            double *flat) {
    int L = n->n_layers;
    double **g = (double **)malloc(sizeof(double *) * (L - 1));
    size_t off = 0;
    for (int l = 0; l < L - 1; l++) {
        g[l] = flat + off;
        off += (size_t)n->dims[l] * n->dims[l + 1];
    }
    weight_gradient(n, input, expected, g);
    free(g);
}

/* gradientStep: SCE:297-346 / GNN:317-366.
 * X is B rows of d_0, Y is B rows of d_{L-1}; rows are visited in the order given (the
 * reference's order is HashMap identity-hash order, i.e. unspecified: SURVEY H3).
 * batchSize = B (SCE:325). */
int oracle_gradient_step(oracle_net *n, const double *X, const double *Y, int B, double step,
                         double momentum, int noise) {
    if (B <= 0) return 1;
    int L = n->n_layers;
    int d0 = n->dims[0], od = n->dims[L - 1];
    double **agg = (double **)calloc(L - 1, sizeof(double *));
    double **single = (double **)calloc(L - 1, sizeof(double *));
    int have_agg = 0;
    for (int s = 0; s < B; s++) {
        for (int l = 0; l < L - 1; l++) { /* fresh double[d_i][d_j] per sample: SCE:240, SCE:264 */
            size_t sz = (size_t)n->dims[l] * n->dims[l + 1];
            if (n->alloc_per_sample || !single[l]) single[l] = (double *)calloc(sz, sizeof(double));
        }
        weight_gradient(n, X + (size_t)s * d0, Y + (size_t)s * od, single); /* SCE:307 */
        if (have_agg) { /* SCE:308-317 */
            for (int l = 0; l < L - 1; l++) {
                size_t sz = (size_t)n->dims[l] * n->dims[l + 1];
                double *a = agg[l];
                const double *g = single[l];
                for (size_t e = 0; e < sz; e++) a[e] += g[e];
                if (n->alloc_per_sample) { free(single[l]); single[l] = NULL; }
            }
        } else { /* SCE:318-320: the first sample's map is adopted as the accumulator */
            for (int l = 0; l < L - 1; l++) { agg[l] = single[l]; single[l] = NULL; }
            have_agg = 1;
        }
    }
    int batchSize = B; /* SCE:325 */
    for (int l = 0; l < L - 1; l++) { /* SCE:327-342 */
        size_t sz = (size_t)n->dims[l] * n->dims[l + 1];
        double *w = n->weights[l];
        double *by = agg[l];
        const double *pv = n->prev[l];
        for (size_t e = 0; e < sz; e++) {
            double adj = step * by[e] / batchSize + momentum * pv[e]; /* SCE:333 */
            if (noise) /* SCE:334-336 (NaN whenever the draw is negative: SURVEY H9) */
                adj += sqrt(jrandom_next_gaussian(&n->random) * step / pow(1 + n->time, momentum));
            w[e] -= adj;  /* SCE:338 */
            by[e] = adj;  /* SCE:339 */
        }
    }
    n->time++; /* SCE:343 */
    for (int l = 0; l < L - 1; l++) { /* SCE:344: previousUpdate = gradientAggregate */
        free(n->prev[l]);
        n->prev[l] = agg[l];
        if (single[l]) free(single[l]);
    }
    free(agg); free(single);
    return 0;
}

/* argmax rule of MT:166-168 / MT:191-193: `>=` so ties resolve to the HIGHEST index;
 * a NaN at index 0 is sticky, a NaN elsewhere is never selected. */
int oracle_argmax(const double *out, int len) {
    int actual = 0;
    for (int i = 0; i < len; i++)
        if (out[i] >= out[actual]) actual = i;
    return actual;
}

/* accessors (extensions: `weights` is private at SCE:15 with no getter) */
void oracle_get_weights(const oracle_net *n, double *flat) {
    size_t off = 0;
    for (int l = 0; l < n->n_layers - 1; l++) {
        size_t sz = (size_t)n->dims[l] * n->dims[l + 1];
        memcpy(flat + off, n->weights[l], sz * sizeof(double));
        off += sz;
    }
}
void oracle_set_weights(oracle_net *n, const double *flat) {
    size_t off = 0;
    for (int l = 0; l < n->n_layers - 1; l++) {
        size_t sz = (size_t)n->dims[l] * n->dims[l + 1];
        memcpy(n->weights[l], flat + off, sz * sizeof(double));
        off += sz;
    }
}
void oracle_get_momentum(const oracle_net *n, double *flat) {
    size_t off = 0;
    for (int l = 0; l < n->n_layers - 1; l++) {
        size_t sz = (size_t)n->dims[l] * n->dims[l + 1];
        memcpy(flat + off, n->prev[l], sz * sizeof(double));
        off += sz;
    }
}
void oracle_set_momentum(oracle_net *n, const double *flat) {
    size_t off = 0;
    for (int l = 0; l < n->n_layers - 1; l++) {
        size_t sz = (size_t)n->dims[l] * n->dims[l + 1];
        memcpy(n->prev[l], flat + off, sz * sizeof(double));
        off += sz;
    }
}

/* input / label encoding: MT:92-101 (pixel/255.0) and MT:112-118 (one-hot 1.0) */
void oracle_encode_image(const uint8_t *raw, int bytes, double *out) {
    for (int i = 0; i < bytes; i++) out[i] = (raw[i] & 0xff) / 255.0;
}
void oracle_encode_label(int label, int n_classes, double *out) {
    for (int i = 0; i < n_classes; i++) out[i] = 0.0;
    out[label] = 1.0;
}

/* epoch sampler without replacement: NNT:143-168.  `sampler` holds indices into the master
 * key list (the reference holds the double[] keys themselves, in HashMap order: unspecified).
 * Returns the number of DISTINCT samples drawn (duplicates across a refill collapse in the
 * reference's HashMap: SURVEY H11), writing them in first-draw order to out_idx. */
typedef struct {
    int32_t *sampler; int size; int master; jrandom random;
} oracle_sampler;

oracle_sampler *oracle_sampler_create(int master_size, int64_t seed) {
    oracle_sampler *s = (oracle_sampler *)calloc(1, sizeof(*s));
    s->master = master_size;
    s->sampler = (int32_t *)malloc(sizeof(int32_t) * master_size);
    for (int i = 0; i < master_size; i++) s->sampler[i] = i; /* refillSampler NNT:164-168 */
    s->size = master_size;
    jrandom_seed(&s->random, seed); /* NNT:42 */
    return s;
}
void oracle_sampler_destroy(oracle_sampler *s) { if (s) { free(s->sampler); free(s); } }

int oracle_sampler_sample(oracle_sampler *s, int batch, int32_t *out_idx) {
    int n = 0;
    for (int i = 0; i < batch; i++) {
        if (s->size == 0) { /* NNT:149-151 */
            for (int k = 0; k < s->master; k++) s->sampler[k] = k;
            s->size = s->master;
        }
        int r = jrandom_next_int_bound(&s->random, s->size); /* NNT:152 */
        int32_t v = s->sampler[r];                            /* NNT:153 */
        memmove(s->sampler + r, s->sampler + r + 1, sizeof(int32_t) * (s->size - r - 1)); /* NNT:154 */
        s->size--;
        int dup = 0;
        for (int k = 0; k < n; k++) if (out_idx[k] == v) { dup = 1; break; } /* HashMap.put NNT:155 */
        if (!dup) out_idx[n++] = v;
    }
    return n;
}

/* Deterministic synthetic inputs for fixtures: U[0,1) from the java.util.Random stream of
 * `seed` (so the committed golden vectors need no bulk input data), optionally thinned to an
 * MNIST-like density: a value is kept when a second draw is < keep (keep >= 1 keeps all). */
void oracle_fill_uniform(int64_t seed, long n, double keep, double *out) {
    jrandom r;
    jrandom_seed(&r, seed);
    for (long i = 0; i < n; i++) {
        double v = jrandom_next_double(&r);
        if (keep < 1.0) {
            double u = jrandom_next_double(&r);
            if (!(u < keep)) v = 0.0;
        }
        out[i] = v;
    }
}
/* one-hot labels: label_i = nextInt(n_classes) of Random(seed) */
void oracle_fill_onehot(int64_t seed, long rows, int n_classes, double *out) {
    jrandom r;
    jrandom_seed(&r, seed);
    for (long i = 0; i < rows; i++) {
        int c = jrandom_next_int_bound(&r, n_classes);
        for (int j = 0; j < n_classes; j++) out[i * n_classes + j] = (j == c) ? 1.0 : 0.0;
    }
}
