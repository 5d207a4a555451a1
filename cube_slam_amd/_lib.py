"""Loader of libcubeslam_hip.so (the C-ABI in include/cubeslam_hip.h).  No CPU fallback: if the library or a HIP
device is missing this raises."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CUBESLAM_LIB") or os.path.join(_HERE, "libcubeslam_hip.so")  # CUBESLAM_LIB: a development build of the same library (profiling variants)
_LIB = None

CS_OK = 0
STATUS = {0: "CS_OK", -1: "CS_ERR_NO_DEVICE", -2: "CS_ERR_BAD_ARG", -3: "CS_ERR_HIP", -4: "CS_ERR_CAPACITY", -5: "CS_ERR_NOMEM"}


class CubeSlamError(RuntimeError):
    pass


def build(force=False):
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc"), "clean"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc")])


def header_version():
    """CS_VERSION of include/cubeslam_hip.h in this tree (None when the header is not there)."""
    try:
        import re
        m = re.search(r"#define\s+CS_VERSION\s+(\d+)", open(os.path.join(os.path.dirname(_HERE), "include", "cubeslam_hip.h")).read())
        return int(m.group(1)) if m else None
    except OSError:
        return None


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise CubeSlamError("libcubeslam_hip.so is not built (run python -c 'import __graft_entry__ as g; g.build()')")
        _LIB = C.CDLL(LIB_PATH)
        _LIB.cs_last_error.restype = C.c_char_p
        want = header_version()
        if want is not None and _LIB.cs_version() != want:  # an ABI-changing header with a stale library (or the reverse) must not get as far as a call
            raise CubeSlamError("libcubeslam_hip.so is version %d, include/cubeslam_hip.h is %d: rebuild" % (_LIB.cs_version(), want))
    return _LIB


def check(ctx, r, what):
    if r != CS_OK:
        msg = lib().cs_last_error(ctx).decode() if ctx else ""
        raise CubeSlamError("%s failed: %s %s" % (what, STATUS.get(r, r), msg))


class Context:
    """cs_ctx wrapper: one HIP device + stream."""

    def __init__(self, device=0, priority=0):
        self._ctx = C.c_void_p()
        r = lib().cs_create_with_priority(int(device), int(priority), C.byref(self._ctx))
        if r != CS_OK:
            raise CubeSlamError("cs_create(device=%d) failed: %s (no HIP device? this package has no CPU path)" % (device, STATUS.get(r, r)))

    @property
    def ptr(self):
        return self._ctx

    def sync(self):
        check(self._ctx, lib().cs_sync(self._ctx), "cs_sync")

    def timing(self, on=True):
        check(self._ctx, lib().cs_timing_enable(self._ctx, int(on)), "cs_timing_enable")

    def timing_reset(self):
        check(self._ctx, lib().cs_timing_reset(self._ctx), "cs_timing_reset")

    def timing_get(self, name):
        ms, n = C.c_double(), C.c_long()
        check(self._ctx, lib().cs_timing_get(self._ctx, name.encode(), C.byref(ms), C.byref(n)), "cs_timing_get")
        return ms.value, n.value

    @staticmethod
    def comm_unique_id():
        """ncclUniqueId (128 bytes): rank 0 creates it, every rank passes the same bytes to comm_init."""
        buf = (C.c_ubyte * 128)()
        r = lib().cs_comm_unique_id(buf)
        if r != CS_OK:
            raise CubeSlamError("cs_comm_unique_id failed: %s" % STATUS.get(r, r))
        return bytes(buf)

    def comm_init(self, rank, world, unique_id):
        """RCCL communicator of this rank on the context's stream (cs_comm_init)."""
        buf = (C.c_ubyte * 128).from_buffer_copy(bytes(unique_id))
        check(self._ctx, lib().cs_comm_init(self._ctx, int(rank), int(world), buf), "cs_comm_init")

    def close(self):
        if self._ctx:
            lib().cs_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
