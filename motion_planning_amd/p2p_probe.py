"""Child process of motion_planning_amd.p2p.setup: first contact with the peers' mailboxes on a throw-away engine.
    python -m motion_planning_amd.p2p_probe WORLD RANK LOCAL_RANK
prints "HANDLE <hex>", reads the WORLD handles (hex, space separated) from stdin, runs the self-test, prints "P2P_OK"."""
import sys


def main():
    world, rank, local_rank = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    from motion_planning_amd.mppi import Engine
    eng = Engine(64, 50, device=local_rank)
    handle = eng.p2p_create(world, rank)
    print("HANDLE " + handle.hex(), flush=True)
    line = sys.stdin.readline().split()
    handles = [bytes.fromhex(h) for h in line]
    if len(handles) != world:
        sys.exit(3)
    eng.p2p_connect(handles=handles)
    eng.p2p_selftest(16)
    print("P2P_OK", flush=True)
    # keep the mailbox mapped until the slowest peer has finished reading it
    sys.stdin.readline()
    eng.close()


if __name__ == "__main__":
    main()
