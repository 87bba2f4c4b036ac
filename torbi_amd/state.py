"""What the host layer remembers about tensors between calls: ONE store, one lifetime rule.

Several pieces of per-tensor knowledge speed up repeated decodes with one model: the look at the transition matrix's
structure (banded or not), the scan depth a time-resident
launch left behind, the log() / device copy of a caller's transition (`core.from_probabilities`), and which
preparation a workspace holds (`decode(reuse_preparation=True)`).  All of it hangs off the tensor OBJECT and its
version counter:

* `notes(tensor)` returns the dict of notes for this object at its current version (a write to the tensor -- a new
  version -- starts an empty dict), or None for tensors without a version counter (created under
  torch.inference_mode(): nothing is remembered about them);
* an entry lives exactly as long as its tensor: a weak reference removes it when the tensor is collected.  Nothing
  is evicted by count and nothing is ever cleared wholesale, so a hundred other matrices passing through cannot take
  a live matrix's measurements with them, and the store cannot outgrow the tensors that exist;
* `reset()` (exported as torbi_amd.reset_path_state) forgets everything, e.g. between benchmark phases.

Identity, not address: a new tensor may reuse a freed tensor's storage address with the same version and shape.
"""
import threading
import weakref
from typing import Optional

_lock = threading.RLock()        # (re-entrant: a weak-reference callback may run, on this thread, while the lock is held)
_entries = {}          # id(tensor) -> [weakref, version, notes dict]


def _version_of(tensor):
    """A tensor's version counter, or None where it has none (inference tensors raise on `_version`)."""
    try:
        return tensor._version
    except RuntimeError:
        return None


def notes(tensor) -> Optional[dict]:
    """The notes kept for `tensor` at its current version (created empty on first use); None when the tensor has no
    version counter."""
    version = _version_of(tensor)
    if version is None:
        return None
    key = id(tensor)
    with _lock:
        entry = _entries.get(key)
        if entry is not None and entry[0]() is tensor:
            if entry[1] != version:
                entry[1], entry[2] = version, {}
            return entry[2]

        def forget(ref, key=key):
            with _lock:
                found = _entries.get(key)
                if found is not None and found[0] is ref:
                    del _entries[key]

        entry = _entries[key] = [weakref.ref(tensor, forget), version, {}]
        return entry[2]


def peek(tensor) -> Optional[dict]:
    """`notes` without creating anything: None unless notes exist for this object at its current version."""
    version = _version_of(tensor)
    with _lock:
        entry = _entries.get(id(tensor))
        if version is None or entry is None or entry[0]() is not tensor or entry[1] != version:
            return None
        return entry[2]


def every(kind: str):
    """The notes of kind `kind` of every live tensor."""
    with _lock:
        return [entry[2][kind] for entry in _entries.values() if kind in entry[2]]


def size() -> int:
    with _lock:
        return len(_entries)


def reset() -> None:
    """Forget everything that has been learnt about tensors (path measurements, structure looks, prepared copies,
    workspace contents)."""
    with _lock:
        _entries.clear()
