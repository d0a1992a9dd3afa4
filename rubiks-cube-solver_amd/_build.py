"""Build identity of the native libraries: a hash of the sources a library is compiled from, embedded at build time
(-DRC_SRC_HASH=..., returned by rc_build_id() / rc_tree_build_id()) and recomputed from the tree at load time.

A library whose embedded id differs from the tree's sources is STALE: __graft_entry__.build() recompiles it (it compares ids, not
mtimes -- touching the .so hides nothing) and _lib.lib() / _tree.tree_lib() refuse to load it.  No torch import here: build() runs
this before anything else is loaded."""
from __future__ import annotations

import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
MARKER = b"rc-build-id:"                     # the id is stored in the binary as "rc-build-id:<16 hex digits>"

HIP_SOURCES = (os.path.join(_HERE, "csrc", "rubikhip.hip"), os.path.join(_HERE, "csrc", "rc_device.h"), os.path.join(_HERE, "csrc", "rc_tables.h"),
               os.path.join(_ROOT, "include", "rubikhip.h"))
TREE_SOURCES = (os.path.join(_HERE, "csrc", "rc_tree.cpp"), os.path.join(_ROOT, "include", "rubiktree.h"))


def source_hash(paths) -> str | None:
    """sha256 over (file name, length, bytes) of every source, first 16 hex digits; None if a source is absent (an installed copy
    without its sources cannot be checked)."""
    h = hashlib.sha256()
    for p in paths:
        if not os.path.exists(p):
            return None
        data = open(p, "rb").read()
        h.update(os.path.basename(p).encode() + b"\0" + str(len(data)).encode() + b"\0" + data)
    return h.hexdigest()[:16]


def embedded_id(lib_path) -> str | None:
    """The id stored in a built library, read from the file's bytes (no dlopen: build() must be able to replace the file)."""
    if not os.path.exists(lib_path):
        return None
    data = open(lib_path, "rb").read()
    i = data.find(MARKER)
    if i < 0:
        return None
    return data[i + len(MARKER):i + len(MARKER) + 16].decode("ascii", "replace")


def check_loaded(name, reported: str, paths):
    """Raise if the id a LOADED library reports is not the hash of the sources in this tree (RC_ALLOW_STALE=1 turns it into a pass:
    A/B experiments that load an older build through RUBIKHIP_LIB)."""
    want = source_hash(paths)
    if want is None or reported == want or os.environ.get("RC_ALLOW_STALE") == "1":
        return
    raise RuntimeError(f"{name} is stale: it was built from sources with id {reported!r}, the tree's sources have id {want!r}; rebuild with "
                       "`python -c 'import __graft_entry__ as g; g.build()'`")
