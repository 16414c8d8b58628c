// Reference-side binding (SURVEY 8f N3): a NeuralNet implementation whose arithmetic runs in
// libgnn_mlp_hip.so.  NOT compiled here: the build image has no JDK (see INTEGRATION.md).
import java.util.*;

/** NeuralNet backed by libgnn_mlp_hip.so (MI355X).  Drop-in for SoftmaxCrossEntropyNeuralNet. */
public class HipSoftmaxCrossEntropyNeuralNet implements NeuralNet, AutoCloseable {
    public enum Activation { LEAKY_RELU, SIGMOID, TANH, RELU, IDENTITY }   // gnn_act values 0..4

    static { System.loadLibrary("gnn_mlp_jni"); }       // the shim below; it links libgnn_mlp_hip.so

    private long handle;                                 // gnn_mlp_t*
    private final int[] layerDims;

    public HipSoftmaxCrossEntropyNeuralNet(int[] layerDims, Activation inner, int maxBatch) {
        this.layerDims = layerDims.clone();
        handle = nativeCreate(layerDims, /*out_kind*/0, inner.ordinal(), 4, 0, /*seed*/1L, /*f32*/0, /*device*/0, maxBatch);
    }

    @Override public double[] propagate(double[] input) {
        double[] out = new double[getOutputDim()];
        nativePropagate(handle, input, 1, out);
        return out;
    }
    @Override public double calculateLoss(double[] input, double[] expected) {
        double[] loss = new double[1];
        nativeLoss(handle, input, expected, 1, loss);
        return loss[0];
    }
    @Override public Map<Integer, double[][]> calculateWeightGradient(double[] input, double[] expected) {
        double[] flat = new double[(int) nativeNumParams(handle)];
        nativeWeightGradient(handle, input, expected, 1, flat);
        Map<Integer, double[][]> g = new HashMap<>();
        int off = 0;
        for (int l = 0; l + 1 < layerDims.length; l++) {
            double[][] m = new double[layerDims[l]][layerDims[l + 1]];
            for (double[] row : m) { System.arraycopy(flat, off, row, 0, row.length); off += row.length; }
            g.put(l, m);
        }
        return g;
    }
    @Override public void gradientStep(Map<double[], double[]> batch, double step, double momentum, boolean noise) {
        int b = batch.size(), din = getInputDim(), dout = getOutputDim(), i = 0;   // batch.size(): SCE:325
        double[] x = new double[b * din], y = new double[b * dout];
        for (Map.Entry<double[], double[]> e : batch.entrySet()) {                 // iteration order, as SCE:305
            System.arraycopy(e.getKey(), 0, x, i * din, din);
            System.arraycopy(e.getValue(), 0, y, i * dout, dout);
            i++;
        }
        nativeGradientStep(handle, x, y, b, step, momentum, noise);
    }
    @Override public int getInputDim()  { return layerDims[0]; }
    @Override public int getOutputDim() { return layerDims[layerDims.length - 1]; }
    @Override public void close() { if (handle != 0) { nativeDestroy(handle); handle = 0; } }

    private static native long nativeCreate(int[] dims, int outKind, int innerAct, int lastAct, int loss,
                                            long seed, int dtype, int device, int maxBatch);
    private static native void nativeDestroy(long h);
    private static native long nativeNumParams(long h);
    private static native void nativePropagate(long h, double[] x, int b, double[] out);
    private static native void nativeLoss(long h, double[] x, double[] y, int b, double[] loss);
    private static native void nativeWeightGradient(long h, double[] x, double[] y, int b, double[] flat);
    private static native void nativeGradientStep(long h, double[] x, double[] y, int b,
                                                  double step, double momentum, boolean noise);
}
