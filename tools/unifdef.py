#!/usr/bin/env python3
"""A small unifdef: resolves preprocessor conditionals whose expression only involves macros with a given fixed value (or fixed as undefined) and leaves every other
conditional untouched.  Used in round 5 to prune the measured-and-dropped A/B switches out of csrc/ (the removed paths live on as tools/probes/r05_pruned_switches.patch).
  python tools/unifdef.py FILE NAME=VALUE ... NAME=undef ...      (rewrites FILE in place; prints what it resolved)
Handles #if / #ifdef / #ifndef / #elif / #else / #endif, `defined(X)`, integer literals and C operators; an `#ifndef X / #define X v / #endif` guard of a fixed macro is
removed together with the comment lines that continue it (lines of the guard that start with the comment column)."""
import re
import sys


def evaluate(expr, fixed):
    """-> int, or None when the expression involves an unknown identifier"""
    expr = re.sub(r"//.*$", "", expr).strip()
    expr = re.sub(r"/\*.*?\*/", "", expr)
    def defined(m):
        name = m.group(1) or m.group(2)
        if name in fixed:
            return "0" if fixed[name] is None else "1"
        return "__UNKNOWN__"
    expr = re.sub(r"defined\s*\(\s*(\w+)\s*\)|defined\s+(\w+)", defined, expr)
    def ident(m):
        name = m.group(0)
        if name in fixed:
            return "0" if fixed[name] is None else str(fixed[name])
        return "__UNKNOWN__"
    expr2 = re.sub(r"\b[A-Za-z_]\w*\b", ident, expr)
    if "__UNKNOWN__" in expr2:
        # short-circuit forms:  KNOWN_FALSE && x  /  KNOWN_TRUE || x  (top level only)
        for op, absorbing in (("&&", 0), ("||", 1)):
            parts = [p.strip() for p in expr.split(op)]
            if len(parts) > 1 and all(p.count("(") == p.count(")") for p in parts):
                vals = [evaluate(p, fixed) for p in parts]
                if any(v is not None and bool(v) == bool(absorbing) for v in vals):
                    return absorbing
        return None
    py = expr2.replace("&&", " and ").replace("||", " or ").replace("!", " not ").replace(" not =", "!=")
    py = re.sub(r"(\d+)[uUlL]+", r"\1", py)
    try:
        return int(eval(py, {"__builtins__": {}}, {}))
    except Exception:
        return None


def process(lines, fixed):
    out = []
    stack = []      # per open conditional: dict(resolved, taken_before, emitting, parent_emitting)
    resolved_sites = 0
    i = 0
    while i < len(lines):
        line = lines[i]
        m = re.match(r"\s*#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)", line)
        emitting = all(f["emit"] for f in stack)
        if not m:
            if emitting:
                out.append(line)
            i += 1
            continue
        kind, rest = m.group(1), m.group(2)
        if kind in ("if", "ifdef", "ifndef"):
            if kind == "if":
                v = evaluate(rest, fixed)
            else:
                name = rest.split()[0] if rest.split() else ""
                v = None
                if name in fixed:
                    # (`#ifndef X / #define X v / #endif` of a macro fixed to a value: false -> the default guard goes; of a macro fixed as undefined: true -> contents stay)
                    v = (fixed[name] is not None) if kind == "ifdef" else (fixed[name] is None)
            if v is None:
                stack.append({"resolved": False, "emit": True, "taken": False})
                if emitting:
                    out.append(line)
            else:
                resolved_sites += 1
                stack.append({"resolved": True, "emit": bool(v), "taken": bool(v)})
        elif kind == "elif":
            f = stack[-1]
            if not f["resolved"]:
                if all(g["emit"] for g in stack[:-1]):
                    out.append(line)
            else:
                if f["taken"]:
                    f["emit"] = False
                else:
                    v = evaluate(rest, fixed)
                    if v is None:          # every earlier branch was resolved false: this #elif becomes the chain's #if, and the chain is unresolved from here on
                        if all(g["emit"] for g in stack[:-1]):
                            out.append(re.sub(r"#(\s*)elif", r"#\1if", line, count=1))
                        f["resolved"] = False; f["emit"] = True
                    else:
                        f["emit"] = bool(v); f["taken"] = bool(v)
        elif kind == "else":
            f = stack[-1]
            if not f["resolved"]:
                if all(g["emit"] for g in stack[:-1]):
                    out.append(line)
            else:
                f["emit"] = not f["taken"]; f["taken"] = True
        else:
            f = stack.pop()
            if not f["resolved"] and all(g["emit"] for g in stack):
                out.append(line)
        i += 1
    return out, resolved_sites


def main():
    path = sys.argv[1]
    fixed = {}
    for a in sys.argv[2:]:
        k, v = a.split("=")
        fixed[k] = None if v == "undef" else int(v)
    lines = open(path).read().split("\n")
    out, n = process(lines, fixed)
    # drop `#define X v` lines of fixed macros that survived outside guards
    out = [l for l in out if not (re.match(r"\s*#\s*define\s+(\w+)\b", l) and re.match(r"\s*#\s*define\s+(\w+)\b", l).group(1) in fixed)]
    open(path, "w").write("\n".join(out))
    print(f"{path}: {n} conditional sites resolved, {len(lines) - len(out)} lines removed")


if __name__ == "__main__":
    main()
