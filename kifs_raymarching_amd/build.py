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


def is_current() -> bool:
    return LIB.exists() and STAMP.exists() and STAMP.read_text().strip() == source_hash()


def build(force: bool = False) -> None:
    if not force and is_current():
        return
    subprocess.run(["make", "-C", str(CSRC), "-B"], check=True)
    STAMP.write_text(source_hash() + "\n")
