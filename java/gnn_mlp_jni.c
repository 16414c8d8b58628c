/* JNI shim for java/HipSoftmaxCrossEntropyNeuralNet.java.  NOT compiled here: no
 * jni.h in the build image; build line in INTEGRATION.md section 3. */
#include <jni.h>
#include "gnn_mlp.h"

static void throw_status(JNIEnv *env, int rc) {
    const char *cls = (rc == GNN_ERR_BAD_ARG) ? "java/lang/IllegalArgumentException" : "java/lang/RuntimeException";
    (*env)->ThrowNew(env, (*env)->FindClass(env, cls), gnn_mlp_last_error());
}

JNIEXPORT jlong JNICALL Java_HipSoftmaxCrossEntropyNeuralNet_nativeCreate(JNIEnv *env, jclass c, jintArray dims,
        jint outKind, jint innerAct, jint lastAct, jint loss, jlong seed, jint dtype, jint device, jint maxBatch) {
    jsize n = (*env)->GetArrayLength(env, dims);
    jint *d = (*env)->GetIntArrayElements(env, dims, NULL);
    gnn_mlp_t *h = NULL;
    int rc = gnn_mlp_create((const int32_t *)d, n, outKind, innerAct, lastAct, loss, seed, dtype, device, maxBatch, &h);
    (*env)->ReleaseIntArrayElements(env, dims, d, JNI_ABORT);
    if (rc) { throw_status(env, rc); return 0; }
    return (jlong)(intptr_t)h;
}

JNIEXPORT void JNICALL Java_HipSoftmaxCrossEntropyNeuralNet_nativeGradientStep(JNIEnv *env, jclass c, jlong h,
        jdoubleArray x, jdoubleArray y, jint b, jdouble step, jdouble momentum, jboolean noise) {
    /* the library copies host->device inside the call and keeps no host pointer, so a critical
       section (no copy of the Java arrays) is safe */
    double *px = (*env)->GetPrimitiveArrayCritical(env, x, NULL);
    double *py = (*env)->GetPrimitiveArrayCritical(env, y, NULL);
    int rc = gnn_mlp_gradient_step((gnn_mlp_t *)(intptr_t)h, px, py, b, step, momentum, noise ? 1 : 0);
    (*env)->ReleasePrimitiveArrayCritical(env, y, py, JNI_ABORT);
    (*env)->ReleasePrimitiveArrayCritical(env, x, px, JNI_ABORT);
    if (rc) throw_status(env, rc);
}
/* nativePropagate / nativeLoss / nativeWeightGradient / nativeDestroy / nativeNumParams follow the
   same three lines: pin, call gnn_mlp_<name>, release (outputs with mode 0 to copy back). */
