// Reference-side binding (SURVEY 8f N3): NeuralNet implementations whose arithmetic runs in
// libgnn_mlp_hip.so on MI355X.  These files sit next to the reference's sources (default package) and use its
// own interfaces: NeuralNet, ActivationFunction, ActivationPrime, LossFunction, LossFunctionPrime.
// UNTESTED SKETCH: the build image has no JDK (no javac, no jni.h), so nothing here has been compiled or
// run; the tested boundary is the C ABI (include/gnn_mlp.h) that these natives call one to one.
import java.util.*;

/** Shared base of HipSoftmaxCrossEntropyNeuralNet / HipGeneralNeuralNet: the NeuralNet interface
 *  (NeuralNet.java:7-67) over the C ABI through gnn_mlp_jni.c. */
public abstract class HipNeuralNet implements NeuralNet, AutoCloseable {

    /** gnn_act values 0..4 -- the closed set of (f, f') pairs the GPU path implements. */
    public enum Activation {
        LEAKY_RELU, SIGMOID, TANH, RELU, IDENTITY;

        /** Recognises one of the known pairs from an arbitrary closure pair by probing it (closures cannot
         *  cross to a GPU: SURVEY H6); anything else is refused rather than silently replaced. */
        public static Activation probe(ActivationFunction f, ActivationPrime fp) {
            final double[] xs = {-2.0, -0.5, 0.0, 0.25, 1.0, 3.0};
            for (Activation a : values()) {
                boolean same = true;
                for (double x : xs) {
                    same &= Math.abs(f.func(x) - a.f(x)) <= 1e-12 && Math.abs(fp.func(x) - a.prime(x)) <= 1e-12;
                }
                if (same) return a;
            }
            throw new IllegalArgumentException("activation pair is not one of " + Arrays.toString(values())
                    + ": use the reference's CPU classes for arbitrary closures");
        }
        double f(double z) {
            switch (this) {
                case LEAKY_RELU: return z > 0 ? z : 0.01 * z;              // MNISTTrainer.java:234
                case SIGMOID: return 1.0 / (1.0 + Math.exp(-z));
                case TANH: return Math.tanh(z);
                case RELU: return z > 0 ? z : 0.0;
                default: return z;
            }
        }
        double prime(double z) {
            switch (this) {
                case LEAKY_RELU: return z <= 0.0 ? 0.01 : 1.0;             // MNISTTrainer.java:235
                case SIGMOID: { double s = f(z); return s * (1.0 - s); }
                case TANH: { double t = Math.tanh(z); return 1.0 - t * t; }
                case RELU: return z > 0 ? 1.0 : 0.0;
                default: return 1.0;
            }
        }
    }
    /** gnn_loss values: the one loss the reference documents besides cross entropy. */
    public enum Loss {
        HALF_SQUARED;                                                       // 0.5 (a - y)^2, derivative a - y
        public static Loss probe(LossFunction l, LossFunctionPrime lp) {
            final double[][] pts = {{0.3, 1.0}, {0.9, 0.0}, {-1.5, 2.0}};
            for (double[] p : pts) {
                double d = p[0] - p[1];
                if (Math.abs(l.loss(p[0], p[1]) - 0.5 * d * d) > 1e-12 || Math.abs(lp.func(p[0], p[1]) - d) > 1e-12)
                    throw new IllegalArgumentException("loss pair is not 0.5*(a-y)^2 / (a-y)");
            }
            return HALF_SQUARED;
        }
    }

    static { System.loadLibrary("gnn_mlp_jni"); }     // gnn_mlp_jni.c; it links libgnn_mlp_hip.so

    protected final int[] layerDims;
    private long dp;       // gnn_mlp_dp_t*: one handle over all devices (0 when single-device)
    private long net;      // gnn_mlp_t*: the (first) replica -- propagate / loss / weights go here

    /** devices.length == 1: plain handle on that GPU; more: gnn_mlp_dp_* with the RCCL reducer. */
    protected HipNeuralNet(int[] layerDims, int outKind, Activation inner, Activation last, Loss loss,
                           int[] devices, int maxBatch, boolean bf16) {
        if (layerDims == null || layerDims.length < 2) throw new IllegalArgumentException("layerDims"); // SCE:105
        this.layerDims = layerDims.clone();
        int dtype = bf16 ? 1 : 0;
        if (devices.length == 1) {
            net = nativeCreate(layerDims, outKind, inner.ordinal(), last.ordinal(), loss.ordinal(), 1L, dtype, devices[0], maxBatch);
        } else {
            dp = nativeCreateDp(layerDims, outKind, inner.ordinal(), last.ordinal(), loss.ordinal(), 1L, dtype, devices, maxBatch, /*RCCL*/0);
            net = nativeReplica(dp, 0);
        }
    }

    @Override public double[] propagate(double[] input) {                               // NeuralNet.java:16
        double[] out = new double[getOutputDim()];
        nativePropagate(net, input, 1, out);
        return out;
    }
    /** Batched form for callers that hold many rows (MNISTTrainer's accuracy loops, MT:159-197). */
    public int[] argmax(double[] rows, int b) {
        int[] labels = new int[b];
        nativeArgmax(net, rows, b, labels);                                             // `>=`: ties -> highest index
        return labels;
    }
    @Override public double calculateLoss(double[] input, double[] expected) {          // NeuralNet.java:27
        double[] loss = new double[1];
        nativeLoss(net, input, expected, 1, loss);
        return loss[0];
    }
    @Override public Map<Integer, double[][]> calculateWeightGradient(double[] input, double[] expected) { // NeuralNet.java:39
        double[] flat = new double[(int) nativeNumParams(net)];
        nativeWeightGradient(net, input, expected, 1, flat);
        return unflatten(flat);
    }
    @Override public void gradientStep(Map<double[], double[]> batch, double step, double momentum, boolean noise) { // NeuralNet.java:51
        int b = batch.size(), din = getInputDim(), dout = getOutputDim(), i = 0;        // batch.size(): SCE:325
        if (b == 0) throw new IllegalArgumentException("empty batch");                  // SCE:300
        double[] x = new double[b * din], y = new double[b * dout];
        for (Map.Entry<double[], double[]> e : batch.entrySet()) {                      // iteration order, as SCE:305
            System.arraycopy(e.getKey(), 0, x, i * din, din);
            System.arraycopy(e.getValue(), 0, y, i * dout, dout);
            i++;
        }
        if (dp != 0) nativeGradientStepDp(dp, x, y, b, step, momentum, noise);          // rows sharded over the GPUs
        else nativeGradientStep(net, x, y, b, step, momentum, noise);
    }
    @Override public int getInputDim()  { return layerDims[0]; }                         // NeuralNet.java:58
    @Override public int getOutputDim() { return layerDims[layerDims.length - 1]; }      // NeuralNet.java:65

    /** Extensions (the reference keeps `weights` private with no accessor, SCE:15). */
    public Map<Integer, double[][]> getWeights() {
        double[] flat = new double[(int) nativeNumParams(net)];
        nativeGetWeights(net, flat);
        return unflatten(flat);
    }
    public int getTime() { return nativeTime(net); }                                     // `time`, SCE:23,343
    public void saveCheckpoint(String path) { nativeSaveCheckpoint(net, path); }
    public void loadCheckpoint(String path) {
        if (dp != 0) throw new IllegalStateException("load into a single-device net, or set the weights on every replica");
        nativeLoadCheckpoint(net, path);
    }

    @Override public void close() {
        if (dp != 0) { nativeDestroyDp(dp); dp = 0; net = 0; }
        else if (net != 0) { nativeDestroy(net); net = 0; }
    }

    private Map<Integer, double[][]> unflatten(double[] flat) {     // layer-major, row-major [in][out] (SCE:44-47)
        Map<Integer, double[][]> g = new HashMap<>();
        int off = 0;
        for (int l = 0; l + 1 < layerDims.length; l++) {
            double[][] m = new double[layerDims[l]][layerDims[l + 1]];
            for (double[] row : m) { System.arraycopy(flat, off, row, 0, row.length); off += row.length; }
            g.put(l, m);
        }
        return g;
    }

    // one native per C entry point (gnn_mlp_jni.c)
    private static native long nativeCreate(int[] dims, int outKind, int innerAct, int lastAct, int loss,
                                            long seed, int dtype, int device, int maxBatch);
    private static native long nativeCreateDp(int[] dims, int outKind, int innerAct, int lastAct, int loss,
                                              long seed, int dtype, int[] devices, int maxBatch, int reducer);
    private static native long nativeReplica(long dp, int r);
    private static native void nativeDestroy(long h);
    private static native void nativeDestroyDp(long dp);
    private static native long nativeNumParams(long h);
    private static native int nativeTime(long h);
    private static native void nativePropagate(long h, double[] x, int b, double[] out);
    private static native void nativeArgmax(long h, double[] x, int b, int[] labels);
    private static native void nativeLoss(long h, double[] x, double[] y, int b, double[] loss);
    private static native void nativeWeightGradient(long h, double[] x, double[] y, int b, double[] flat);
    private static native void nativeGradientStep(long h, double[] x, double[] y, int b, double step, double momentum, boolean noise);
    private static native void nativeGradientStepDp(long dp, double[] x, double[] y, int b, double step, double momentum, boolean noise);
    private static native void nativeGetWeights(long h, double[] flat);
    private static native void nativeSaveCheckpoint(long h, String path);
    private static native void nativeLoadCheckpoint(long h, String path);
}
