"""One rank of tests/test_multirank.py::test_rendezvous_survives_an_abandoned_connection: the solver's own rendezvous
(pangulu_amd_comm_init: TCP control plane), a barrier, finalize.  No solver, no GPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pangulu_amd import _lib  # noqa: E402


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    lib = _lib.load("r64", test_hooks=True)
    assert lib.pangulu_amd_comm_init(rank, world, b"127.0.0.1", port, _lib.TRANSPORT_HOST, None) == 0
    lib.pangulu_amd_comm_barrier()
    lib.pangulu_amd_comm_finalize()
    print("rank %d ok" % rank, flush=True)


if __name__ == "__main__":
    main()
