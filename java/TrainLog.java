// UNTESTED SKETCH (no JDK in the build image) -- see HipNeuralNet.java.
import java.io.*;
import java.util.Locale;

/** The row MNISTTrainer.logTest appends to logs/trainLog.csv (MNISTTrainer.java:211-219), for a trainer
 *  that drives a Hip* net: `dims,iterations,step,batch,momentum,noise,trainAcc,testAcc` with the reference's
 *  number formats (`784-300-100-10,100000,0.01250,16,0.900,false,0.99973,0.97800`, logs/trainLog.csv:4),
 *  and the observer line of NeuralNetTrainer.train (`"%d,%.2f\n"`, NeuralNetTrainer.java:71). */
public final class TrainLog {
    private TrainLog() {}

    public static String row(int[] netDim, int iterations, double stepSize, int batchSize, double momentum,
                             boolean noise, double trainingAcc, double testAcc) {
        StringBuilder sb = new StringBuilder();
        sb.append(netDim[0]);
        for (int i = 1; i < netDim.length; i++) sb.append('-').append(netDim[i]);
        sb.append(String.format(Locale.ROOT, ",%d,%.5f,%d,%.3f,", iterations, stepSize, batchSize, momentum));
        sb.append(noise).append(',');
        sb.append(String.format(Locale.ROOT, "%.5f,%.5f\n", trainingAcc, testAcc));
        return sb.toString();
    }

    /** Appends one row, creating logs/ like the reference's FileOutputStream(..., true) (MNISTTrainer.java:61). */
    public static void append(File csv, int[] netDim, int iterations, double stepSize, int batchSize,
                              double momentum, boolean noise, double trainingAcc, double testAcc) throws IOException {
        File dir = csv.getParentFile();
        if (dir != null) dir.mkdirs();
        try (PrintStream out = new PrintStream(new FileOutputStream(csv, true))) {
            out.print(row(netDim, iterations, stepSize, batchSize, momentum, noise, trainingAcc, testAcc));
        }
    }

    public static String observerLine(int iteration, double validationLoss) {
        return String.format(Locale.ROOT, "%d,%.2f\n", iteration, validationLoss);
    }
}
