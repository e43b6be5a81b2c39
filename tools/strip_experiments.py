"""Remove experiment-only preprocessor branches from the product kernels: every macro in UNDEF is
treated as undefined, `#ifdef / #ifndef / #if defined(X) [&& ...] / #elif defined(X) / #else / #endif` blocks on them are
resolved and the dead branch deleted; everything else is left untouched.  The variants removed this way (rotated chunk
loop, register epilogue of the chain kernels, three-stage conv2 phase, burst / spread builds, cycle probes, ablations)
are documented with their measurements in profiles/r03_probes.md / r04_probes.md and live on in the git history
(last commit with them: see profiles/r04_probes.md).

    python tools/strip_experiments.py [--check] file..."""
import re
import sys

UNDEF = {"FCP_CHAIN_ROT", "FCP_CHAIN_REG_EPI", "FCP_CHAIN_C2_STAGES3", "FCP_CHAIN_BURST", "FCP_CHAIN_SPREAD", "FCP_CHAIN_NOALIAS",
         "FCP_CHAIN_PROBE", "FCP_BIG_PROBE", "FCP_BIG_ABLATE", "FCP_HALO_PROBE", "FCP_STEM_PROBE", "FCP_STEM_ABLATE_STAGE",
         "FCP_STEM_ABLATE_AREAD", "FCP_NO_FMA_MIX", "FCP_WIDE2_BT", "FCP_CONV_PROFILING", "FCP_BIG_STAGGER", "FCP_HALO_ABLATE"}

COND = re.compile(r"^\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)$")


def evaluate(kind, rest):
    """-> True / False when the condition is decided by UNDEF alone, None when it is none of our business."""
    rest = rest.split("//")[0].strip()
    if kind in ("ifdef", "ifndef"):
        name = rest.split()[0]
        if name not in UNDEF:
            return None
        return kind == "ifndef"
    names = re.findall(r"defined\s*\(\s*(\w+)\s*\)", rest)
    if not names or not any(n in UNDEF for n in names):
        return None
    if "||" in rest and not all(n in UNDEF for n in names):
        return None
    # `defined(X) && ...` with X undefined is false; `defined(X) || defined(Y)` with both undefined is false
    if rest.lstrip().startswith("!"):
        return None
    return False


def strip(text):
    out, stack = [], []          # stack entries: [ours, emitting_before, taken_already, currently_emitting]
    for line in text.split("\n"):
        m = COND.match(line)
        emitting = all(s[3] for s in stack)
        if not m:
            if emitting:
                out.append(line)
            continue
        kind, rest = m.group(1), m.group(2)
        if kind in ("ifdef", "ifndef", "if"):
            val = evaluate(kind, rest) if emitting else None
            if val is None:
                stack.append([False, emitting, True, True])
                if emitting:
                    out.append(line)
            else:
                stack.append([True, emitting, val, val])
        elif kind == "elif":
            top = stack[-1]
            if not top[0]:
                if all(s[3] for s in stack[:-1]):
                    out.append(line)
                continue
            val = evaluate("if", rest)
            if val is None:
                raise SystemExit(f"cannot resolve: {line}")
            top[3] = (not top[2]) and val
            top[2] = top[2] or val
        elif kind == "else":
            top = stack[-1]
            if not top[0]:
                if all(s[3] for s in stack[:-1]):
                    out.append(line)
                continue
            top[3] = not top[2]
            top[2] = True
        else:
            top = stack.pop()
            if not top[0] and all(s[3] for s in stack):
                out.append(line)
    assert not stack, "unbalanced conditionals"
    return "\n".join(out)


if __name__ == "__main__":
    check = "--check" in sys.argv
    bad = 0
    for path in [a for a in sys.argv[1:] if not a.startswith("--")]:
        src = open(path).read()
        new = strip(src)
        if new != src:
            bad += 1
            print(("would change " if check else "stripped ") + path, len(src.split("\n")), "->", len(new.split("\n")), "lines")
            if not check:
                open(path, "w").write(new)
    sys.exit(1 if check and bad else 0)
