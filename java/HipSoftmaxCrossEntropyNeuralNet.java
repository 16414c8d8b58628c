// UNTESTED SKETCH (no JDK in the build image) -- see HipNeuralNet.java.

/** Drop-in for SoftmaxCrossEntropyNeuralNet (SoftmaxCrossEntropyNeuralNet.java:103): configurable inner
 *  activation, softmax output, cross-entropy loss; Random(1) initial weights reproduced exactly. */
public class HipSoftmaxCrossEntropyNeuralNet extends HipNeuralNet {

    /** The reference's constructor signature: the closure pair is recognised by probing (leaky ReLU is what
     *  MNISTTrainer.java:234-235 passes), so `new SoftmaxCrossEntropyNeuralNet(dims, f, fp)` becomes
     *  `new HipSoftmaxCrossEntropyNeuralNet(dims, f, fp)` and nothing else changes. */
    public HipSoftmaxCrossEntropyNeuralNet(int[] layerDims, ActivationFunction innerActivationFunc,
                                           ActivationPrime innerActivationPrime) {
        this(layerDims, Activation.probe(innerActivationFunc, innerActivationPrime), new int[]{0}, 1024, false);
    }

    /** devices: one entry = that GPU; several = the batch sharded over them inside the library (RCCL). */
    public HipSoftmaxCrossEntropyNeuralNet(int[] layerDims, Activation inner, int[] devices, int maxBatch, boolean bf16) {
        super(layerDims, /*GNN_OUT_SOFTMAX_CE*/0, inner, Activation.IDENTITY, Loss.HALF_SQUARED, devices, maxBatch, bf16);
    }
}
