"""Builds libkifs_hip.so in-tree with hipcc for gfx950 and records a hash of the sources,
so that a stale library (sources edited, library not rebuilt) is detected, not used."""
import hashlib
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libkifs_hip.so"
STAMP = PKG / "libkifs_hip.so.srchash"
HEADER = PKG.parent / "include" / "kifs_hip.h"


def source_hash() -> str:
    h = hashlib.sha256()
    files = sorted(p for p in CSRC.iterdir()
                   if p.suffix in (".hip", ".cpp", ".hpp", ".h") or p.name == "Makefile")
    for p in files + [HEADER]:
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.hexdigest()


def library_hash() -> str:
    return hashlib.sha256(LIB.read_bytes()).hexdigest() if LIB.exists() else ""


def write_stamp() -> None:
    """Two lines: the hash of the sources the library was built from, and the hash of the library file itself --
    so that a library copied over the built one (a sweep's variant) is noticed as well as an edited source."""
    STAMP.write_text(source_hash() + "\n" + library_hash() + "\n")


def recorded() -> tuple:
    """(source hash, library hash) of the stamp; ("", "") without one."""
    if not STAMP.exists():
        return "", ""
    lines = STAMP.read_text().split()
    return (lines[0] if lines else ""), (lines[1] if len(lines) > 1 else "")


def is_current() -> bool:
    src, lib = recorded()
    return LIB.exists() and src == source_hash() and lib == library_hash()


def build(force: bool = False) -> None:
    if not force and is_current():
        return
    subprocess.run(["make", "-C", str(CSRC), "-B"], check=True)
    write_stamp()
