"""Helpers of the import-path shim: find and run the REFERENCE's file of the same name further down a package path.

`dropin/<pkg>/__init__.py` extends `<pkg>.__path__` over every `<pkg>/` directory on sys.path (pkgutil.extend_path), so
submodules this directory does not provide keep resolving to the reference checkout.  Two things that mechanism cannot do
are done here: running the reference package's own `__init__.py` (datasets/__init__.py defines `dataset_dict`,
utils/__init__.py the optimizer helpers -- train.py:9,22 import them from the PACKAGE), and loading the reference's
version of a module the shim overrides (so that the shim can subclass / delegate to it)."""
import importlib.util
import os
import sys


def reference_dirs(pkg_path, own_dir):
    own = os.path.realpath(own_dir)
    return [d for d in pkg_path if os.path.realpath(d) != own]


def run_reference_init(pkg_globals, own_dir):
    """exec the first other `__init__.py` on the package path in the shim package's namespace"""
    for d in reference_dirs(pkg_globals["__path__"], own_dir):
        f = os.path.join(d, "__init__.py")
        if os.path.isfile(f):
            with open(f) as fh:
                exec(compile(fh.read(), f, "exec"), pkg_globals)
            return f
    return None


def load_reference_module(pkg_name, mod_name, own_dir):
    """the reference's `<pkg>/<mod>.py` as module `<pkg>._reference_<mod>` (None when no checkout is on the path)"""
    alias = "%s._reference_%s" % (pkg_name, mod_name)
    if alias in sys.modules:
        return sys.modules[alias]
    for d in reference_dirs(sys.modules[pkg_name].__path__, own_dir):
        f = os.path.join(d, mod_name + ".py")
        if os.path.isfile(f):
            spec = importlib.util.spec_from_file_location(alias, f)
            mod = importlib.util.module_from_spec(spec)
            sys.modules[alias] = mod
            spec.loader.exec_module(mod)
            return mod
    return None
