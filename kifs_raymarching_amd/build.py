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


def kernel_hash() -> str:
    """Hash of what decides the DEVICE code and the launch shapes: the kernel files, every header, the scheduling rules
    (kifs_schedule.cpp) and the Makefile's flags -- what a profile's counters and an instruction count belong to.
    Host-only edits (kifs_api / kifs_multi / kifs_shards / kifs_host) leave it alone."""
    h = hashlib.sha256()
    files = sorted(p for p in CSRC.iterdir()
                   if p.suffix in (".hip", ".hpp", ".h") or p.name in ("Makefile", "kifs_schedule.cpp"))
    for p in files:
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.hexdigest()


def library_hash() -> str:
    return hashlib.sha256(LIB.read_bytes()).hexdigest() if LIB.exists() else ""


def write_stamp() -> None:
    """Three lines: the hash of the sources the library was built from, the hash of the library file itself -- so that
    a library copied over the built one (a sweep's variant) is noticed as well as an edited source -- and the
    kernel hash (kernel_hash(): what profiles/pmc_traffic.json and profiles/lone_frame_floor.json entries are tied to)."""
    STAMP.write_text(source_hash() + "\n" + library_hash() + "\n" + kernel_hash() + "\n")


def recorded() -> tuple:
    """(source hash, library hash, kernel hash) of the stamp; empty strings without one."""
    lines = STAMP.read_text().split() if STAMP.exists() else []
    return tuple(lines[i] if len(lines) > i else "" for i in range(3))


def is_current() -> bool:
    src, lib, _ = recorded()
    return LIB.exists() and src == source_hash() and lib == library_hash()


def build(force: bool = False) -> None:
    if not force and is_current():
        if recorded()[2] != kernel_hash():  # (a stamp from before the kernel hash existed)
            write_stamp()
        return
    subprocess.run(["make", "-C", str(CSRC), "-B"], check=True)
    write_stamp()
