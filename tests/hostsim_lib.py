"""Binds the kernel-logic simulator (tests/hostsim/_build/libcdbg_hostsim.so).
TEST INFRASTRUCTURE: same C ABI, same kernel source, workgroup threads as CPU fibers."""
import os
import subprocess

from bcalm_amd import api

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "hostsim", "_build", "libcdbg_hostsim.so")
SRC = os.path.join(os.path.dirname(HERE), "bcalm_amd", "csrc")
HOST = os.path.join(os.path.dirname(HERE), "bcalm_amd", "host")           # (the simulator build of the `bcalm` CLI is made by the same script)


def load():
    newest = max([os.path.getmtime(os.path.join(SRC, f)) for f in os.listdir(SRC)] + [os.path.getmtime(os.path.join(HOST, f)) for f in os.listdir(HOST)] +
                 [os.path.getmtime(os.path.join(HERE, "hostsim", f)) for f in ("hostsim.h", "build.sh")])
    if not os.path.exists(SO) or os.path.getmtime(SO) < newest:
        subprocess.check_call([os.path.join(HERE, "hostsim", "build.sh")], stdout=subprocess.DEVNULL)
    return api.load(SO)
