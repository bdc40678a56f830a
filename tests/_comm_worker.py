"""Rank process of tests/test_comm_rendezvous.py: host-only communicator (device -1, no HIP / RCCL call),
`rounds` shared-memory exchanges with rank- and round-dependent payloads, then a barrier."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "g-vom_amd")]
import gvom_sharded  # noqa: E402


def main():
    rank, world, name, rounds, delay_ms = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    die_at = int(sys.argv[6]) if len(sys.argv) > 6 else -1      # the LAST rank leaves without a word before this round
    if delay_ms and rank == 0:
        import time
        time.sleep(delay_ms / 1000.0)                  # the others are already polling for the segment
    comm = gvom_sharded.RcclComm(rank, world, -1, name)
    for i in range(rounds):
        k = 1 + (i % 5)
        if i == die_at and rank == world - 1:
            os._exit(7)
        try:
            got = comm.exchange_host([rank * 1000003 + i * 7 + j for j in range(k)])
        except Exception as e:                         # (only with die_at: the library noticed that a rank is gone)
            print("round %d: %s" % (i, e))
            sys.exit(5)
        want = [[r * 1000003 + i * 7 + j for j in range(k)] for r in range(world)]
        if got != want:
            print("rank %d round %d: %r != %r" % (rank, i, got, want))
            sys.exit(3)
    comm.barrier()
    try:
        comm.allgather_rows(type("B", (), {"h": None})())
        sys.exit(4)                                    # device collectives must be refused on a host-only communicator
    except Exception:
        pass
    comm.close()
    print("ok %d" % rank)


if __name__ == "__main__":
    main()
