// UNTESTED SKETCH (no JDK in the build image) -- see HipNeuralNet.java.

/** Drop-in for GeneralNeuralNet (GeneralNeuralNet.java:112-115): inner activation, output activation and
 *  loss are each one of the closed enums; output delta = loss'(f_last(z), y) * f_last'(z) with the
 *  PRE-activation z (GeneralNeuralNet.java:268-271). */
public class HipGeneralNeuralNet extends HipNeuralNet {

    /** The reference's six-closure constructor; each pair is recognised by probing. */
    public HipGeneralNeuralNet(int[] layerDims,
                               ActivationFunction innerActivationFunc, ActivationPrime innerActivationPrime,
                               ActivationFunction lastActivationFunc, ActivationPrime lastActivationPrime,
                               LossFunction lossFunc, LossFunctionPrime lossPrime) {
        this(layerDims, Activation.probe(innerActivationFunc, innerActivationPrime),
             Activation.probe(lastActivationFunc, lastActivationPrime), Loss.probe(lossFunc, lossPrime),
             new int[]{0}, 1024, false);
    }

    public HipGeneralNeuralNet(int[] layerDims, Activation inner, Activation last, Loss loss,
                               int[] devices, int maxBatch, boolean bf16) {
        super(layerDims, /*GNN_OUT_ACT_LOSS*/1, inner, last, loss, devices, maxBatch, bf16);
    }
}
