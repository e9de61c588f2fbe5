"""kernel_source_hash of bench.py for the tools that write profiles/*.json: the counter files are stamped with the hash of the kernel sources they were measured on"""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash() -> str:
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    return mod.kernel_source_hash()
