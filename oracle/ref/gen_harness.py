"""Generate the Fortran registration harness for an oracle/_ref build.

TEST INFRASTRUCTURE ONLY (see oracle/README.md).  This script reads the
*preprocessed* reference headers (the output of `cpp -traditional -P` over
/root/reference/**/*.h, produced by oracle/build_ref.py into a scratch
directory) and emits a fixed-form Fortran file whose only content is

    include "<reference header>"
    call orc_reg('<name>', <variable>, <elem bytes>, <typecode>, rank, shape, lbound)

for every variable that lives in a COMMON block.  The emitted file therefore
contains no reference text: only `include` lines and one call per variable
name.  `orc_reg_` is implemented in oracle/ref/harness.c and stores the base
address, so that the Python driver (oracle/refmodel.py) can view every COMMON
array of the compiled reference as a numpy array.
"""
from __future__ import annotations

import re
import sys
from pathlib import Path

# headers whose COMMON blocks the drivers need (names relative to inc/)
HEADERS = [
    "mw.h", "isopyc.h", "grdvar.h", "coord.h", "levind.h", "vmixc.h",
    "hmixc.h", "accel.h", "scalar.h", "switch.h", "state.h", "tmngr.h",
    "csbc.h", "diag.h", "diaga.h", "ice.h", "atm.h", "cembm.h", "mobi.h", "emode.h",
    "cregin.h", "timeavgs.h", "index.h", "cfilt.h", "calendar.h", "tidal_kv.h", "cpolar.h",
]

TYPECODE = {"real": 1, "integer": 2, "logical": 3}


def logical_lines(text: str):
    """Join fixed-form continuation lines; drop comments."""
    out = []
    for raw in text.splitlines():
        if not raw.strip():
            continue
        if raw[0] in "cC*!":
            continue
        line = raw.split("!")[0].rstrip() if "'" not in raw else raw.rstrip()
        if not line.strip():
            continue
        if len(line) > 5 and line[5] not in " 0" and line[:5].strip() == "":
            if out:
                out[-1] += " " + line[6:].strip()
            continue
        out.append(line.strip())
    return out


def split_top(s: str):
    """Split on commas that are not inside parentheses."""
    parts, depth, cur = [], 0, ""
    for ch in s:
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur.strip())
    return parts


def parse_header(path: Path):
    """Return (types, commons): name->type and ordered list of (block, name, dims)."""
    types: dict[str, str] = {}
    commons: list[tuple[str, str, str | None]] = []
    dims_from_decl: dict[str, str] = {}
    for line in logical_lines(path.read_text(encoding="latin-1")):
        low = line.lower()
        m = re.match(r"(real|integer|logical)\b(\s*\*\s*\d+)?\s*(::)?\s*(.*)$", low)
        if m and not low.startswith(("real function", "integer function", "logical function")):
            typ, rest = m.group(1), line[m.start(4):]
            if "=" in rest and "(" not in rest.split("=")[0]:
                # statement function or initialisation, ignore
                pass
            for item in split_top(rest):
                mm = re.match(r"([A-Za-z_]\w*)\s*(\((.*)\))?$", item.strip())
                if mm:
                    types[mm.group(1).lower()] = typ
                    if mm.group(3):
                        dims_from_decl[mm.group(1).lower()] = mm.group(3)
            continue
        m = re.match(r"common\s*/\s*(\w+)\s*/\s*(.*)$", line, flags=re.I)
        if m:
            block = m.group(1).lower()
            for item in split_top(m.group(2)):
                mm = re.match(r"([A-Za-z_]\w*)\s*(\((.*)\))?$", item.strip())
                if mm:
                    commons.append((block, mm.group(1), mm.group(3)))
    # attach dims given on the type declaration (e.g. "real Ahh(km)")
    commons = [(b, n, d if d else dims_from_decl.get(n.lower())) for b, n, d in commons]
    return types, commons


def emit(inc_dir: Path, out: Path):
    lines = []
    subs = []
    for h in HEADERS:
        p = inc_dir / h
        if not p.exists():
            continue
        types, commons = parse_header(p)
        regs = [(b, n, d) for (b, n, d) in commons if types.get(n.lower()) in TYPECODE]
        if not regs:
            continue
        sub = "orc_reg_" + h.replace(".h", "")
        subs.append(sub)
        L = lines.append
        L(f"      subroutine {sub}")
        L("      implicit none")
        L("      integer i, k, j, ip, kr, jq, n, jp, jrow, orc_idum(1)")
        L('      include "size.h"')
        L('      include "param.h"')
        L('      include "pconst.h"')
        L('      include "stdunits.h"')
        extra = {"isopyc.h": [], "mobi.h": [], "diaga.h": [], "timeavgs.h": [],
                 "cfilt.h": [], "index.h": []}.get(h, [])
        for e in extra:
            L(f'      include "{e}"')
        if h not in ("size.h", "param.h", "pconst.h", "stdunits.h"):
            L(f'      include "{h}"')
        L("      orc_idum(1) = 0")
        for b, n, d in regs:
            tc = TYPECODE[types[n.lower()]]
            if d:
                L(f"      call orc_reg('{b}:{n.lower()}', {n},")
                L(f"     &  storage_size({n})/8, {tc}, size(shape({n})),")
                L(f"     &  shape({n}), lbound({n}))")
            else:
                L(f"      call orc_reg('{b}:{n.lower()}', {n},")
                L(f"     &  storage_size({n})/8, {tc}, 0, orc_idum, orc_idum)")
        L("      return")
        L("      end")
        L("")
    lines += ["      subroutine orc_flush", "      flush(6)", "      return", "      end", ""]
    lines.append("      subroutine orc_register_all")
    lines.append("      implicit none")
    for s in subs:
        lines.append(f"      call {s}")
    lines.append("      return")
    lines.append("      end")
    out.write_text("\n".join(lines) + "\n")


if __name__ == "__main__":
    emit(Path(sys.argv[1]), Path(sys.argv[2]))
