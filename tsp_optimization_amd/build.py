"""Builds the native libraries in-tree (hipcc for gfx950, gcc for the C host mirror)."""
import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.environ.get("TSP_LIB_DIR") or os.path.join(PKG, "lib")   # TSP_LIB_DIR: a diagnostic or A/B build (tools/)


def lib_path(name="libtsp_hip.so"):
    return os.path.join(LIB_DIR, name)


def build_hip(verbose=False):
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = ["make", "-C", os.path.join(PKG, "csrc"), "-j8"]
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return lib_path()


def build_host(verbose=False):
    host = os.path.join(PKG, "host")
    if not os.path.exists(os.path.join(host, "Makefile")):
        return None
    subprocess.check_call(["make", "-C", host, "-j8"], stdout=None if verbose else subprocess.DEVNULL)
    return lib_path("libtsp_host.so")


def build_all(verbose=False):
    build_hip(verbose)
    build_host(verbose)
