"""Build helper of the librccl test double (fake_rccl.cpp): TEST INFRASTRUCTURE."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def build(no_hip=False):
    """-> path of the built library.  no_hip: the socket protocol alone on host memory (CPU tests); else `librccl.so.1`, the name
    stx_comm.cpp dlopens, linked against the HIP runtime (GPU tests put this directory in front of LD_LIBRARY_PATH)."""
    src = os.path.join(HERE, "fake_rccl.cpp")
    out = os.path.join(HERE, "libfake_rccl_nohip.so" if no_hip else "librccl.so.1")
    if os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    cmd = ["g++", "-O1", "-shared", "-fPIC", "-std=c++17", src, "-o", out, "-lpthread"]
    if no_hip:
        cmd.insert(1, "-DFAKE_RCCL_NO_HIP")
    else:
        cmd += ["-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-L/opt/rocm/lib", "-lamdhip64"]
    subprocess.check_call(cmd)
    return out
