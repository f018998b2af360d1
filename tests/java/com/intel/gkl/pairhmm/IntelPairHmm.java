// For the first machine that has BOTH a JDK and an MI355X (none of this project's boxes has a JDK: docs/NOTES.md 52).
// Not compiled or run by any test here.  A stand-in for GKL's com.intel.gkl.pairhmm.IntelPairHmm with the same three
// native methods (reference src/main/java/com/intel/gkl/pairhmm/IntelPairHmm.java:157-166) -- the class name decides the
// JNI symbol names, so it must be this one -- that loads libgkl_pairhmm.so and replays the reference's golden file
// (tests/golden/pairhmm-testdata.txt: one read x one haplotype per call, absolute tolerance 1e-5, like
// PairHmmUnitTest.dataFileTest, src/test/java/com/intel/gkl/pairhmm/PairHmmUnitTest.java:171-234), then one
// 300 x 24 batch against itself in double precision.
//
//   javac -d /tmp/gklhip-classes tests/java/com/intel/gkl/pairhmm/IntelPairHmm.java
//   java -Xcheck:jni -cp /tmp/gklhip-classes com.intel.gkl.pairhmm.IntelPairHmm \
//        $PWD/gkl_amd/lib/libgkl_pairhmm.so tests/golden/pairhmm-testdata.txt
//
// -Xcheck:jni is what the reference's own test JVMs run with (build.gradle:101-104); the only warnings it may print are
// on malformed input (a quality array shorter than readBases), which this program does not send.
package com.intel.gkl.pairhmm;

import java.io.BufferedReader;
import java.io.FileReader;
import java.util.ArrayList;
import java.util.List;
import java.util.Random;

public class IntelPairHmm {
    // the fields JavaData.h:55-62 reads reflectively by name, all byte[]
    public static class ReadDataHolder {
        public byte[] readBases, readQuals, insertionGOP, deletionGOP, overallGCP;
    }

    public static class HaplotypeDataHolder {
        public byte[] haplotypeBases;
    }

    private static native void initNative(Class<?> readDataHolderClass, Class<?> haplotypeDataHolderClass,
                                          boolean doublePrecision, int maxThreads);

    private native void computeLikelihoodsNative(Object[] readDataArray, Object[] haplotypeDataArray, double[] likelihoodArray);

    private native void doneNative();

    private static byte[] quals(String s, int floor) {   // ASCII-33, clamped like PairHmmUnitTest.java:211-214
        byte[] q = s.getBytes();
        for (int i = 0; i < q.length; i++) q[i] = (byte) Math.max(q[i] - 33, floor);
        return q;
    }

    public static void main(String[] args) throws Exception {
        System.load(args[0]);
        final IntelPairHmm hmm = new IntelPairHmm();
        int cases = 0, bad = 0;
        double worst = 0;
        for (int pass = 0; pass < 2; pass++) {
            initNative(ReadDataHolder.class, HaplotypeDataHolder.class, pass == 1, pass == 0 ? 1 : 4);
            try (BufferedReader in = new BufferedReader(new FileReader(args[1]))) {
                String line;
                while ((line = in.readLine()) != null) {
                    if (line.startsWith("#") || line.trim().isEmpty()) continue;
                    final String[] c = line.trim().split("\\s+");
                    final HaplotypeDataHolder h = new HaplotypeDataHolder();
                    h.haplotypeBases = c[0].getBytes();
                    final ReadDataHolder r = new ReadDataHolder();
                    r.readBases = c[1].getBytes();
                    r.readQuals = quals(c[2], 6);
                    r.insertionGOP = quals(c[3], 0);
                    r.deletionGOP = quals(c[4], 0);
                    r.overallGCP = quals(c[5], 0);
                    final double[] out = new double[1];
                    hmm.computeLikelihoodsNative(new Object[]{r}, new Object[]{h}, out);
                    final double err = Math.abs(out[0] - Double.parseDouble(c[6]));
                    worst = Math.max(worst, err);
                    cases++;
                    if (!(err <= 1e-5)) bad++;
                }
            }
        }
        // a batch big enough for the pipelined path when GKL_HIP_JNI_PIPELINE_PAIRS=1 is set: float and double agree to 1e-5 relative
        final Random rng = new Random(20250418);
        final byte[] alphabet = "ACGT".getBytes();
        final List<HaplotypeDataHolder> haps = new ArrayList<>();
        final byte[] window = new byte[400];
        for (int i = 0; i < window.length; i++) window[i] = alphabet[rng.nextInt(4)];
        for (int k = 0; k < 24; k++) {
            final HaplotypeDataHolder h = new HaplotypeDataHolder();
            h.haplotypeBases = window.clone();
            for (int e = 0; e < 3; e++) h.haplotypeBases[rng.nextInt(window.length)] = alphabet[rng.nextInt(4)];
            haps.add(h);
        }
        final List<ReadDataHolder> reads = new ArrayList<>();
        for (int k = 0; k < 300; k++) {
            final int len = 50 + rng.nextInt(200), off = rng.nextInt(window.length - len);
            final ReadDataHolder r = new ReadDataHolder();
            r.readBases = java.util.Arrays.copyOfRange(window, off, off + len);
            r.readQuals = new byte[len];
            r.insertionGOP = new byte[len];
            r.deletionGOP = new byte[len];
            r.overallGCP = new byte[len];
            for (int i = 0; i < len; i++) {
                r.readQuals[i] = (byte) (6 + rng.nextInt(35));
                r.insertionGOP[i] = (byte) (30 + rng.nextInt(16));
                r.deletionGOP[i] = (byte) (30 + rng.nextInt(16));
                r.overallGCP[i] = 10;
            }
            reads.add(r);
        }
        final double[] f = new double[300 * 24], d = new double[300 * 24];
        initNative(ReadDataHolder.class, HaplotypeDataHolder.class, false, 4);
        hmm.computeLikelihoodsNative(reads.toArray(), haps.toArray(), f);
        initNative(ReadDataHolder.class, HaplotypeDataHolder.class, true, 4);
        hmm.computeLikelihoodsNative(reads.toArray(), haps.toArray(), d);
        double rel = 0;
        for (int i = 0; i < f.length; i++) rel = Math.max(rel, Math.abs(f[i] - d[i]) / Math.abs(d[i]));
        hmm.doneNative();
        System.out.printf("golden cases (float + double): %d, outside 1e-5: %d, worst |error| %.3g; 300 x 24 batch float vs double: max relative difference %.3g%n",
                          cases, bad, worst, rel);
        System.exit(bad == 0 && rel < 1e-5 ? 0 : 1);
    }
}
