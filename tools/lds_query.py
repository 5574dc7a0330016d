"""Prints the LDS limits of device 0 as the HIP runtime reports them (diagnostic)."""
import ctypes as C
hip = C.CDLL("libamdhip64.so")
v = C.c_int(0)
# hipDeviceAttribute_t values (hip_runtime_api.h): MaxSharedMemoryPerBlock = 43 in the CUDA-compatible range, look a few up by name via hipDeviceGetAttribute
names = {"hipDeviceAttributeMaxSharedMemoryPerBlock": None, "hipDeviceAttributeMaxSharedMemoryPerMultiprocessor": None}
# enumerate a window of attribute ids and print the ones that look like LDS sizes
for a in range(0, 120):
    if hip.hipDeviceGetAttribute(C.byref(v), a, 0) == 0 and v.value in (65536, 163840, 49152, 98304, 131072, 160 * 1024):
        print("attr", a, "=", v.value)
