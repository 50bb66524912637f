#!/usr/bin/env python3
"""Undefined-name check without third-party linters (none are installed here): every name a function of the given files reads
as a global must be defined at module level (or be a builtin).  The benchmark code only runs on a GPU box - a misspelt name
should not cost a lease.  Usage: python tools/check_names.py file.py ..."""
import builtins
import symtable
import sys

IMPLICIT = {"__file__", "__name__", "__doc__", "__package__"}


def walk(tab, module_names, path, errors):
    for sym in tab.get_symbols():
        if tab.get_type() == "function" and sym.is_global() and sym.is_referenced():
            if sym.get_name() not in module_names and sym.get_name() not in IMPLICIT and not hasattr(builtins, sym.get_name()):
                errors.append("%s: %s() reads undefined global %r" % (path, tab.get_name(), sym.get_name()))
    for child in tab.get_children():
        walk(child, module_names, path, errors)


def main():
    errors = []
    for path in sys.argv[1:]:
        src = open(path).read()
        top = symtable.symtable(src, path, "exec")
        names = {s.get_name() for s in top.get_symbols() if s.is_assigned() or s.is_imported() or s.is_namespace()}
        for sym in top.get_symbols():  # module-level reads
            if sym.is_referenced() and sym.get_name() not in names and sym.get_name() not in IMPLICIT and not hasattr(builtins, sym.get_name()):
                errors.append("%s: module reads undefined name %r" % (path, sym.get_name()))
        walk(top, names, path, errors)
    print("\n".join(errors) if errors else "ok: %d file(s)" % len(sys.argv[1:]))
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main())
