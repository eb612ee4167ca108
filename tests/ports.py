"""A rendezvous port for ranks started by a test.  Asking the kernel for an ephemeral port (bind to 0), closing it and handing
the number to the ranks races with every outgoing connection of the box -- they draw from the same range -- and a rank then
dies with EADDRINUSE (seen once in 300 GPU cases).  Ports below the ephemeral range are only ever taken by listeners."""
import random
import socket


def free_port():
    lo = 20000
    try:
        hi = min(int(open("/proc/sys/net/ipv4/ip_local_port_range").read().split()[0]), 32768)
    except (OSError, ValueError, IndexError):
        hi = 32768
    if hi - lo < 1000:
        lo, hi = 10000, 20000
    for _ in range(200):
        p = random.randrange(lo, hi)
        s = socket.socket()
        try:
            s.bind(("127.0.0.1", p))
        except OSError:
            continue
        finally:
            s.close()
        return p
    raise RuntimeError("no free rendezvous port")
