"""Worker process of the streaming executor's polish helpers (`xmris_amd.autophase_solver.PolishWorkers`).

A search whose best member does not pass scipy's projected-gradient test is polished on the reference's own route --
scipy's L-BFGS-B on the NUMPY objective (processing/phasing.py:276-284 with scipy's defaults): a few milliseconds of
small numpy operations, i.e. of interpreter.  Run on threads of the process that queues the kernels, those milliseconds
are fought for under its interpreter lock (heterogeneous datasets: 13 searches of 16 need the polish, and the launch
thread fell from 1.4 to 2.3 ms per dataset); run here, they cost that process nothing.  The worker never touches the
GPU or the HIP library: it imports numpy, scipy and the objective statements only, and talks length-prefixed pickles
over its stdin / stdout."""
import pickle
import struct
import sys


def main():
    from xmris_amd import autophase_solver as aps

    try:  # (now, while nobody waits: the first request must not pay for scipy's import)
        import scipy.optimize  # noqa: F401
        from scipy.optimize import _lbfgsb  # noqa: F401
    except ImportError:
        pass
    inp, out = sys.stdin.buffer, sys.stdout.buffer
    sys.stdout = sys.stderr  # (anything a library prints must not end up in the reply stream)
    out.write(b"XMREADY\n")  # (eight bytes: the parent hands requests only to workers that have said this)
    out.flush()
    while True:
        head = inp.read(8)
        if len(head) < 8:
            return
        args = pickle.loads(inp.read(struct.unpack("<q", head)[0]))
        try:
            reply = ("ok", aps.polish_reference(*args))
        except Exception as e:  # noqa: BLE001 -- reported to the caller, which polishes itself
            reply = ("error", repr(e))
        blob = pickle.dumps(reply, protocol=pickle.HIGHEST_PROTOCOL)
        out.write(struct.pack("<q", len(blob)))
        out.write(blob)
        out.flush()


if __name__ == "__main__":
    main()
