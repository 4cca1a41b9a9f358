"""TEST INFRASTRUCTURE: builds tests/simt/_build/libmlp_emu.so -- the kernels of on-policy_amd/csrc/mappo_mlp_impl.h
compiled for the host against the SIMT emulator (simt_emu.h) with ROCm's clang++ (ext_vector_type support)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "_build", "libmlp_emu.so")
SOURCES = [os.path.join(HERE, "mlp_emu.cc"), os.path.join(HERE, "simt_emu.h"),
           os.path.join(ROOT, "on-policy_amd", "csrc", "mappo_mlp_impl.h"),
           os.path.join(ROOT, "on-policy_amd", "csrc", "mappo_gru_impl.h"),
           os.path.join(ROOT, "on-policy_amd", "csrc", "mappo_lin_impl.h"), os.path.join(ROOT, "include", "mappo_hip.h")]
CLANG = os.environ.get("MAPPO_HOST_CLANG", "/opt/rocm/lib/llvm/bin/clang++")


def _fresh():
    return os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in SOURCES)


def build():
    if _fresh():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    # one builder at a time (pytest-xdist workers reach this together): the others wait and find the library fresh; the
    # library appears under its name only when complete
    import fcntl
    with open(OUT + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not _fresh():
            tmp = OUT + ".tmp.%d" % os.getpid()
            subprocess.run([CLANG, "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-psabi",
                            "-I" + os.path.join(ROOT, "include"), SOURCES[0], "-o", tmp], check=True)
            os.replace(tmp, OUT)
    return OUT


if __name__ == "__main__":
    print(build())
