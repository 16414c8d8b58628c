/* JNI shim for java/HipNeuralNet.java: one native per C entry point of include/gnn_mlp.h.
 * UNTESTED SKETCH: the build image has no jni.h; build line in INTEGRATION.md section 3.
 *
 * Arrays cross by COPY (Get/Set<Type>ArrayRegion into malloc'd buffers), never by
 * GetPrimitiveArrayCritical: the library calls below copy host->device from pageable memory, launch kernels
 * and may run a ~0.4 s hiprtc compile (the 16th step of a handle) -- blocking work that the JNI
 * specification forbids inside a critical region. */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include "gnn_mlp.h"

static void throw_status(JNIEnv *env, int rc) {
    const char *cls = (rc == GNN_ERR_BAD_ARG) ? "java/lang/IllegalArgumentException"
                    : (rc == GNN_ERR_UNSUPPORTED) ? "java/lang/UnsupportedOperationException"
                    : (rc == GNN_ERR_STATE) ? "java/lang/IllegalStateException" : "java/lang/RuntimeException";
    (*env)->ThrowNew(env, (*env)->FindClass(env, cls), gnn_mlp_last_error());
}
static void throw_oom(JNIEnv *env) { (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/OutOfMemoryError"), "gnn_mlp_jni: malloc"); }

/* a copy of a Java double[] / int[] (caller frees); NULL + pending exception on failure */
static double *copy_doubles(JNIEnv *env, jdoubleArray a, jsize *n_out) {
    if (!a) { (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/NullPointerException"), "array"); return NULL; }
    jsize n = (*env)->GetArrayLength(env, a);
    double *p = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
    if (!p) { throw_oom(env); return NULL; }
    (*env)->GetDoubleArrayRegion(env, a, 0, n, p);
    if (n_out) *n_out = n;
    return p;
}
static jint *copy_ints(JNIEnv *env, jintArray a, jsize *n_out) {
    if (!a) { (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/NullPointerException"), "array"); return NULL; }
    jsize n = (*env)->GetArrayLength(env, a);
    jint *p = (jint *)malloc(sizeof(jint) * (size_t)(n ? n : 1));
    if (!p) { throw_oom(env); return NULL; }
    (*env)->GetIntArrayRegion(env, a, 0, n, p);
    if (n_out) *n_out = n;
    return p;
}
/* Array lengths are checked HERE against what the C call will touch (the library sees plain pointers): a short array
 * is an IllegalArgumentException, as the reference's asserts / ArrayIndexOutOfBounds would be, never a heap overrun. */
static int len_ok(JNIEnv *env, jarray a, long long need, const char *what) {
    if (!a) { (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/NullPointerException"), what); return 0; }
    if (need < 0 || (long long)(*env)->GetArrayLength(env, a) != need) {
        (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/IllegalArgumentException"), what);
        return 0;
    }
    return 1;
}
static int rows_ok(JNIEnv *env, gnn_mlp_t *h, jint b, jdoubleArray x, jdoubleArray y) {
    if (!h || b <= 0) { (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/IllegalArgumentException"), "handle / batch"); return 0; }
    if (!len_ok(env, x, (long long)b * gnn_mlp_input_dim(h), "rows: length must be b * getInputDim()")) return 0;
    if (y && !len_ok(env, y, (long long)b * gnn_mlp_output_dim(h), "expected: length must be b * getOutputDim()")) return 0;
    return 1;
}
#define H(h) ((gnn_mlp_t *)(intptr_t)(h))
#define DP(h) ((gnn_mlp_dp_t *)(intptr_t)(h))

JNIEXPORT jlong JNICALL Java_HipNeuralNet_nativeCreate(JNIEnv *env, jclass c, jintArray dims, jint outKind, jint innerAct,
        jint lastAct, jint loss, jlong seed, jint dtype, jint device, jint maxBatch) {
    jsize n = 0;
    jint *d = copy_ints(env, dims, &n);
    if (!d) return 0;
    gnn_mlp_t *h = NULL;
    int rc = gnn_mlp_create((const int32_t *)d, n, outKind, innerAct, lastAct, loss, seed, dtype, device, maxBatch, &h);
    free(d);
    if (rc) { throw_status(env, rc); return 0; }
    return (jlong)(intptr_t)h;
}

JNIEXPORT jlong JNICALL Java_HipNeuralNet_nativeCreateDp(JNIEnv *env, jclass c, jintArray dims, jint outKind, jint innerAct,
        jint lastAct, jint loss, jlong seed, jint dtype, jintArray devices, jint maxBatch, jint reducer) {
    jsize n = 0, nd = 0;
    jint *d = copy_ints(env, dims, &n);
    if (!d) return 0;
    jint *dv = copy_ints(env, devices, &nd);
    if (!dv) { free(d); return 0; }
    gnn_mlp_dp_t *h = NULL;
    int rc = gnn_mlp_dp_create((const int32_t *)d, n, outKind, innerAct, lastAct, loss, seed, dtype, (const int32_t *)dv, nd,
                               maxBatch, reducer, &h);
    free(d); free(dv);
    if (rc) { throw_status(env, rc); return 0; }
    return (jlong)(intptr_t)h;
}

JNIEXPORT jlong JNICALL Java_HipNeuralNet_nativeReplica(JNIEnv *env, jclass c, jlong dp, jint r) {
    gnn_mlp_t *h = NULL;
    int rc = gnn_mlp_dp_replica(DP(dp), r, &h);
    if (rc) { throw_status(env, rc); return 0; }
    return (jlong)(intptr_t)h;
}

JNIEXPORT void JNICALL Java_HipNeuralNet_nativeDestroy(JNIEnv *env, jclass c, jlong h) { (void)gnn_mlp_destroy(H(h)); }
JNIEXPORT void JNICALL Java_HipNeuralNet_nativeDestroyDp(JNIEnv *env, jclass c, jlong dp) { (void)gnn_mlp_dp_destroy(DP(dp)); }
JNIEXPORT jlong JNICALL Java_HipNeuralNet_nativeNumParams(JNIEnv *env, jclass c, jlong h) { return (jlong)gnn_mlp_num_params(H(h)); }
JNIEXPORT jint JNICALL Java_HipNeuralNet_nativeTime(JNIEnv *env, jclass c, jlong h) { return (jint)gnn_mlp_time(H(h)); }

JNIEXPORT void JNICALL Java_HipNeuralNet_nativePropagate(JNIEnv *env, jclass c, jlong h, jdoubleArray x, jint b, jdoubleArray out) {
    if (!rows_ok(env, H(h), b, x, NULL) || !len_ok(env, out, (long long)b * gnn_mlp_output_dim(H(h)), "out: length must be b * getOutputDim()")) return;
    double *px = copy_doubles(env, x, NULL);
    if (!px) return;
    jsize no = (*env)->GetArrayLength(env, out);
    double *po = (double *)malloc(sizeof(double) * (size_t)(no ? no : 1));
    if (!po) { free(px); throw_oom(env); return; }
    int rc = gnn_mlp_propagate(H(h), px, b, po);
    if (!rc) (*env)->SetDoubleArrayRegion(env, out, 0, no, po);
    free(px); free(po);
    if (rc) throw_status(env, rc);
}

JNIEXPORT void JNICALL Java_HipNeuralNet_nativeArgmax(JNIEnv *env, jclass c, jlong h, jdoubleArray x, jint b, jintArray labels) {
    if (!rows_ok(env, H(h), b, x, NULL) || !len_ok(env, labels, b, "labels: length must be b")) return;
    double *px = copy_doubles(env, x, NULL);
    if (!px) return;
    int32_t *pl = (int32_t *)malloc(sizeof(int32_t) * (size_t)(b > 0 ? b : 1));
    if (!pl) { free(px); throw_oom(env); return; }
    int rc = gnn_mlp_argmax(H(h), px, b, pl);
    if (!rc) (*env)->SetIntArrayRegion(env, labels, 0, b, (const jint *)pl);
    free(px); free(pl);
    if (rc) throw_status(env, rc);
}

JNIEXPORT void JNICALL Java_HipNeuralNet_nativeLoss(JNIEnv *env, jclass c, jlong h, jdoubleArray x, jdoubleArray y, jint b, jdoubleArray loss) {
    if (!rows_ok(env, H(h), b, x, y) || !len_ok(env, loss, b, "loss: length must be b")) return;
    double *px = copy_doubles(env, x, NULL);
    if (!px) return;
    double *py = copy_doubles(env, y, NULL);
    if (!py) { free(px); return; }
    double *pl = (double *)malloc(sizeof(double) * (size_t)(b > 0 ? b : 1));
    if (!pl) { free(px); free(py); throw_oom(env); return; }
    int rc = gnn_mlp_loss(H(h), px, py, b, pl);
    if (!rc) (*env)->SetDoubleArrayRegion(env, loss, 0, b, pl);
    free(px); free(py); free(pl);
    if (rc) throw_status(env, rc);
}

JNIEXPORT void JNICALL Java_HipNeuralNet_nativeWeightGradient(JNIEnv *env, jclass c, jlong h, jdoubleArray x, jdoubleArray y, jint b, jdoubleArray flat) {
    if (!rows_ok(env, H(h), b, x, y)) return;
    double *px = copy_doubles(env, x, NULL);
    if (!px) return;
    double *py = copy_doubles(env, y, NULL);
    if (!py) { free(px); return; }
    jsize nf = (*env)->GetArrayLength(env, flat);
    double *pf = (double *)malloc(sizeof(double) * (size_t)(nf ? nf : 1));
    if (!pf) { free(px); free(py); throw_oom(env); return; }
    int rc = (nf == (jsize)gnn_mlp_num_params(H(h))) ? gnn_mlp_weight_gradient(H(h), px, py, b, pf) : GNN_ERR_BAD_ARG;
    if (!rc) (*env)->SetDoubleArrayRegion(env, flat, 0, nf, pf);
    free(px); free(py); free(pf);
    if (rc) throw_status(env, rc);
}

JNIEXPORT void JNICALL Java_HipNeuralNet_nativeGradientStep(JNIEnv *env, jclass c, jlong h, jdoubleArray x, jdoubleArray y, jint b,
        jdouble step, jdouble momentum, jboolean noise) {
    if (!rows_ok(env, H(h), b, x, y)) return;
    double *px = copy_doubles(env, x, NULL);
    if (!px) return;
    double *py = copy_doubles(env, y, NULL);
    if (!py) { free(px); return; }
    /* the call returns once the host rows have been consumed (rounded into the library's pinned slot), so the copies can go */
    int rc = gnn_mlp_gradient_step(H(h), px, py, b, step, momentum, noise ? 1 : 0);
    free(px); free(py);
    if (rc) throw_status(env, rc);
}

JNIEXPORT void JNICALL Java_HipNeuralNet_nativeGradientStepDp(JNIEnv *env, jclass c, jlong dp, jdoubleArray x, jdoubleArray y, jint b,
        jdouble step, jdouble momentum, jboolean noise) {
    gnn_mlp_t *r0 = NULL;
    if (gnn_mlp_dp_replica(DP(dp), 0, &r0)) { throw_status(env, GNN_ERR_BAD_ARG); return; }
    if (!rows_ok(env, r0, b, x, y)) return;
    double *px = copy_doubles(env, x, NULL);
    if (!px) return;
    double *py = copy_doubles(env, y, NULL);
    if (!py) { free(px); return; }
    int rc = gnn_mlp_dp_gradient_step(DP(dp), px, py, b, step, momentum, noise ? 1 : 0);
    free(px); free(py);
    if (rc) throw_status(env, rc);
}

JNIEXPORT void JNICALL Java_HipNeuralNet_nativeGetWeights(JNIEnv *env, jclass c, jlong h, jdoubleArray flat) {
    jsize nf = (*env)->GetArrayLength(env, flat);
    double *pf = (double *)malloc(sizeof(double) * (size_t)(nf ? nf : 1));
    if (!pf) { throw_oom(env); return; }
    int rc = (nf == (jsize)gnn_mlp_num_params(H(h))) ? gnn_mlp_get_weights(H(h), pf) : GNN_ERR_BAD_ARG;
    if (!rc) (*env)->SetDoubleArrayRegion(env, flat, 0, nf, pf);
    free(pf);
    if (rc) throw_status(env, rc);
}

JNIEXPORT void JNICALL Java_HipNeuralNet_nativeSaveCheckpoint(JNIEnv *env, jclass c, jlong h, jstring path) {
    const char *p = (*env)->GetStringUTFChars(env, path, NULL);
    if (!p) return;
    int rc = gnn_mlp_save_checkpoint(H(h), p);
    (*env)->ReleaseStringUTFChars(env, path, p);
    if (rc) throw_status(env, rc);
}

JNIEXPORT void JNICALL Java_HipNeuralNet_nativeLoadCheckpoint(JNIEnv *env, jclass c, jlong h, jstring path) {
    const char *p = (*env)->GetStringUTFChars(env, path, NULL);
    if (!p) return;
    int rc = gnn_mlp_load_checkpoint(H(h), p);
    (*env)->ReleaseStringUTFChars(env, path, p);
    if (rc) throw_status(env, rc);
}
