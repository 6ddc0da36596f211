"""Tracer tables of the reference's option sets.

The order of the prognostic tracers is the one `tracer_init` establishes
(/root/reference/updates/09/source/common/UVic_ESCM.F:1281-1370), the order of
the source-term slots is the one of the `itrc` map (same file, :1376-1483) and
the order of the MOBI column tracers is the one of `mobi_init`'s `setimobi`
calls (/root/reference/updates/09/source/mom/mobi.F:440-504).  Index arrays are
integer data and are carried bit-exact (tests/test_configs.py compares them
with the compiled reference when oracle/_ref is present).
"""
from __future__ import annotations

from dataclasses import dataclass, field


@dataclass(frozen=True)
class OptionSet:
    name: str
    tracers: tuple          # prognostic tracer names, 1-based order of the reference
    sources: tuple          # names of tracers that own a source slot, slot order
    mobi: tuple             # names of MOBI column tracers (tnpzd), column order
    options: frozenset = field(default_factory=frozenset)

    @property
    def nt(self) -> int:
        return len(self.tracers)

    @property
    def nsrc(self) -> int:
        return len(self.sources)

    @property
    def ntnpzd(self) -> int:
        return len(self.mobi)

    def index(self, name: str) -> int:
        """1-based tracer index (0 when the tracer is not in this set)."""
        return self.tracers.index(name) + 1 if name in self.tracers else 0

    def itrc(self):
        """1-based source slot of each tracer, 0 = no source (reference `itrc`)."""
        return [self.sources.index(t) + 1 if t in self.sources else 0 for t in self.tracers]

    def imobi(self, name: str) -> int:
        return self.mobi.index(name) + 1 if name in self.mobi else 0


def _build(name, opts):
    o = frozenset(opts)
    has = o.__contains__
    tr = ["temp", "salt"]
    if has("carbon"):
        tr.append("dic")
        if has("carbon_13"):
            tr.append("dic13")
        if has("carbon_14"):
            tr.append("c14")
    if has("mobi_alk"):
        tr.append("alk")
    if has("mobi_o2"):
        tr.append("o2")
    if has("mobi"):
        tr += ["po4", "phyt", "phyt_phos", "zoop", "detr", "detr_phos"]
        if has("mobi_caco3"):
            tr.append("caco3")
        if has("mobi_silicon"):
            tr += ["diat", "sil", "opl"]
        if has("mobi_nitrogen"):
            tr += ["dop", "no3", "don", "diaz"]
            if has("mobi_nitrogen_15"):
                tr += ["din15", "don15", "phytn15"]
                if has("mobi_silicon"):
                    tr.append("diatn15")
                tr += ["zoopn15", "detrn15", "diazn15"]
        if has("mobi_iron"):
            tr += ["dfe", "detrfe"]
        if has("carbon_13"):
            tr.append("phytc13")
            if has("mobi_silicon"):
                tr.append("diatc13")
            if has("mobi_caco3"):
                tr.append("caco3c13")
            tr += ["zoopc13", "detrc13"]
            if has("mobi_nitrogen"):
                tr += ["doc13", "diazc13"]
    src = []
    if has("carbon") and has("carbon_14"):
        src.append("c14")
    if has("mobi"):
        if has("carbon"):
            src.append("dic")
            if has("carbon_13"):
                src.append("dic13")
        if has("mobi_alk"):
            src.append("alk")
        if has("mobi_o2"):
            src.append("o2")
        src += ["po4", "phyt", "phyt_phos", "zoop", "detr", "detr_phos"]
        if has("mobi_iron"):
            src += ["dfe", "detrfe"]
        if has("mobi_caco3"):
            src.append("caco3")
        if has("mobi_silicon"):
            src += ["diat", "sil", "opl"]
        if has("mobi_nitrogen"):
            src += ["dop", "no3", "don", "diaz"]
            if has("mobi_nitrogen_15"):
                src += ["din15", "don15", "phytn15"]
                if has("mobi_silicon"):
                    src.append("diatn15")
                src += ["zoopn15", "detrn15", "diazn15"]
        if has("carbon_13"):
            src.append("phytc13")
            if has("mobi_silicon"):
                src.append("diatc13")
            if has("mobi_caco3"):
                src.append("caco3c13")
            src += ["zoopc13", "detrc13"]
            if has("mobi_nitrogen"):
                src += ["doc13", "diazc13"]
    mobi = []      # column order of tnpzd: imobi* as tracer_init assigns them (checked against the compiled reference)
    if has("mobi"):
        mobi += ["po4", "phyt", "phyt_phos", "zoop", "detr", "detr_phos"]
        if has("carbon"):
            mobi.append("dic")
        if has("carbon_13"):
            mobi += ["dic13", "phytc13", "zoopc13", "detrc13"]
            if has("mobi_nitrogen"):
                mobi += ["doc13", "diazc13"]
            if has("mobi_silicon"):
                mobi.append("diatc13")
            if has("mobi_caco3"):
                mobi.append("caco3c13")
        if has("mobi_nitrogen"):
            mobi += ["dop", "no3", "don", "diaz"]
            if has("mobi_nitrogen_15"):
                mobi += ["din15", "don15", "phytn15", "zoopn15", "detrn15", "diazn15"]
                if has("mobi_silicon"):
                    mobi.append("diatn15")
        if has("mobi_caco3"):
            mobi.append("caco3")
        if has("mobi_silicon"):
            mobi += ["diat", "sil", "opl"]
        if has("mobi_iron"):
            mobi += ["dfe", "detrfe"]
    return OptionSet(name, tuple(tr), tuple(src), tuple(mobi), o)


OPTION_SETS = {
    # BASELINE config 1: physics only
    "p2": _build("p2", []),
    # the same two tracers; names the oracle/_ref build that also holds the momentum routines (`clinic`, §8f rank 4)
    "m2": _build("m2", []),
    "m2i": _build("m2i", []),
    # BASELINE config 4 == SURVEY option set C (nt=30, nsrc=28, ntnpzd=25)
    "c30": _build("c30", ["mobi", "mobi_o2", "mobi_iron", "carbon", "mobi_alk", "mobi_nitrogen",
                          "carbon_13", "carbon_14", "mobi_nitrogen_15"]),
    # the same set; names the oracle/_ref build with the momentum routines and O_time_step_monitor (the tsiperts steps of
    # the shipped run/control.in: tests/test_fortran_shim.py)
    "t30": _build("t30", ["mobi", "mobi_o2", "mobi_iron", "carbon", "mobi_alk", "mobi_nitrogen",
                          "carbon_13", "carbon_14", "mobi_nitrogen_15"]),
    # SURVEY option set E (nt=13): the nearest buildable set to BASELINE config 2
    "e13": _build("e13", ["mobi", "mobi_o2", "mobi_iron", "carbon", "mobi_caco3"]),
    # SURVEY option set F (nt=18): the nearest buildable set to BASELINE config 3
    "f18": _build("f18", ["mobi", "mobi_o2", "mobi_iron", "carbon", "mobi_caco3", "mobi_alk", "mobi_nitrogen"]),
    # the shipped run/mk.in set (nt=37): C + prognostic CaCO3 + diatoms/silicon
    "s37": _build("s37", ["mobi", "mobi_o2", "mobi_iron", "carbon", "mobi_alk", "mobi_nitrogen", "carbon_13", "carbon_14",
                          "mobi_nitrogen_15", "mobi_caco3", "mobi_silicon"]),
}


def performance_set(nt: int) -> OptionSet:
    """Transport-only performance shape with `nt` tracers (BASELINE configs 2/3:
    nt=8/15 cannot be built from the reference): T, S and nt-2
    passive tracers that each own a source slot."""
    names = ("temp", "salt") + tuple(f"trc{n:02d}" for n in range(3, nt + 1))
    return OptionSet(f"perf{nt}", names, names[2:], (), frozenset())
