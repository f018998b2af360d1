"""Exception classes mirroring the Java exceptions the reference surface raises.

``IntelPairHmm.computeLikelihoods`` declares ``NullPointerException,
OutOfMemoryError, IllegalArgumentException`` (reference
src/main/java/com/intel/gkl/pairhmm/IntelPairHmm.java:130-145); the JNI layer
throws the latter two by class path (src/main/native/pairhmm/JavaData.h:130,140,150).
HIP failures surface as ``RuntimeException`` (java/lang/RuntimeException in the
JNI shim).
"""


class NullPointerException(Exception):
    pass


class IllegalArgumentException(ValueError):
    pass


class OutOfMemoryError(MemoryError):
    pass


class RuntimeException(RuntimeError):
    pass
