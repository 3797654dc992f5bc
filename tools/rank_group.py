"""A process group without torch: the ranks of one node agree on wall-clock time (barrier, MAX over ranks), add up
small reports and pass a few bytes around (the 128-byte RCCL communicator id) over a plain TCP socket on 127.0.0.1.

bench.py and its config-4 leg use it for everything that is not the data path: RanSlice.step has no collective,
and the one collective of the build (the shared-dictionary exchange) is ncclAllGather inside libranslice.so.

Topology: rank 0 listens on an ephemeral port and publishes (port, token) in a rendezvous file; the other ranks poll the
file, connect and present the token.  Every operation is an all-gather of one JSON value through rank 0.  The file's name
comes from RANSLICE_RDZV_FILE, or from (MASTER_PORT, parent pid): the ranks of one launch -- torch.distributed.run's
workers, or the children bench.py spawns itself -- share both, a later launch shares neither."""
import json
import os
import secrets
import socket
import struct
import time


def rdzv_dir():
    """A directory only this user can enter: $XDG_RUNTIME_DIR, else /tmp/ranslice-<uid> (created 0700; refused if somebody else
    owns it or others may write to it).  The rendezvous file carries the join token, so it must not sit world-readable under a
    predictable name in /tmp."""
    d = os.environ.get('XDG_RUNTIME_DIR')
    if d and os.path.isdir(d) and os.access(d, os.W_OK | os.X_OK):
        return d
    d = os.path.join('/tmp', 'ranslice-%d' % os.getuid())
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    import stat
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o022):
        raise PermissionError('rank group: %s is not a private directory of this user' % d)
    return d


def default_rdzv_file():
    f = os.environ.get('RANSLICE_RDZV_FILE')
    if f:
        return f
    return os.path.join(rdzv_dir(), 'rdzv_%s_%d' % (os.environ.get('MASTER_PORT', '0'), os.getppid()))


def _publish(path, obj):
    """write `obj` to `path` for the other ranks: a fresh 0600 file (O_EXCL | O_NOFOLLOW: no symlink is followed, nobody's
    pre-created file is reused) moved into place; a stale file of a crashed launch with the same name is removed first"""
    try:
        os.unlink(path)
    except FileNotFoundError:
        pass
    tmp = '%s.%d.%s.tmp' % (path, os.getpid(), secrets.token_hex(4))
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | os.O_NOFOLLOW, 0o600)
    with os.fdopen(fd, 'w') as f:
        json.dump(obj, f)
    os.replace(tmp, path)


def _read_published(path):
    fd = os.open(path, os.O_RDONLY | os.O_NOFOLLOW)
    with os.fdopen(fd) as f:
        if os.fstat(f.fileno()).st_uid != os.getuid():
            raise PermissionError('rank group: %s belongs to another user' % path)
        return json.load(f)


def _send(sock, obj):
    data = json.dumps(obj).encode()
    sock.sendall(struct.pack('<I', len(data)) + data)


def _recv_exact(sock, n):
    buf = b''
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError('rank group: peer closed the connection')
        buf += chunk
    return buf


def _recv(sock):
    (n,) = struct.unpack('<I', _recv_exact(sock, 4))
    return json.loads(_recv_exact(sock, n).decode())


class RankGroup:
    def __init__(self, rank, world, rdzv_file=None, timeout=600.0):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        self.file = rdzv_file or default_rdzv_file()
        self.peers = {}
        self.sock = None
        if self.world == 1:
            return
        if self.rank == 0:
            self._serve()
        else:
            self._join()

    # ---- rendezvous
    def _serve(self):
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind(('127.0.0.1', 0))
        srv.listen(self.world)
        token = secrets.token_hex(16)
        _publish(self.file, {'port': srv.getsockname()[1], 'token': token})
        deadline = time.time() + self.timeout
        while len(self.peers) < self.world - 1:
            srv.settimeout(max(0.1, deadline - time.time()))
            try:
                c, _ = srv.accept()
            except socket.timeout:
                raise TimeoutError('rank group: %d of %d ranks joined within %.0f s'
                                   % (len(self.peers) + 1, self.world, self.timeout))
            c.settimeout(self.timeout)
            try:
                hello = _recv(c)
            except Exception:
                c.close()
                continue
            r = hello.get('rank')
            if hello.get('token') != token or not isinstance(r, int) or not (0 < r < self.world) or r in self.peers:
                c.close()
                continue
            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            _send(c, {'ok': token})
            self.peers[r] = c
        srv.close()

    def _join(self):
        deadline = time.time() + self.timeout
        while True:
            if time.time() > deadline:
                raise TimeoutError('rank group: rank %d found no rank 0 through %s' % (self.rank, self.file))
            try:
                info = _read_published(self.file)
                s = socket.create_connection(('127.0.0.1', int(info['port'])), timeout=2.0)
                s.settimeout(self.timeout)
                _send(s, {'token': info['token'], 'rank': self.rank})
                if _recv(s).get('ok') != info['token']:
                    raise ConnectionError('not the group of this launch')
                s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                self.sock = s
                return
            except (OSError, ValueError, KeyError, ConnectionError):
                time.sleep(0.05)

    # ---- operations
    def allgather(self, value):
        """the values of all ranks, in rank order (value: anything json encodes)"""
        if self.world == 1:
            return [value]
        if self.rank == 0:
            vals = [value] + [None] * (self.world - 1)
            for r, c in self.peers.items():
                vals[r] = _recv(c)
            for c in self.peers.values():
                _send(c, vals)
            return vals
        _send(self.sock, value)
        return _recv(self.sock)

    def barrier(self):
        self.allgather(None)

    def max(self, x):
        return max(float(v) for v in self.allgather(float(x)))

    def sum(self, values):
        rows = self.allgather([float(v) for v in values])
        return [sum(col) for col in zip(*rows)]

    def bcast_bytes(self, data, src=0):
        vals = self.allgather(data.hex() if (self.rank == src and data is not None) else None)
        return bytes.fromhex(vals[src])

    def close(self):
        for c in self.peers.values():
            try:
                c.close()
            except Exception:
                pass
        self.peers = {}
        if self.sock is not None:
            try:
                self.sock.close()
            except Exception:
                pass
            self.sock = None
        if self.world > 1 and self.rank == 0:
            try:
                os.remove(self.file)
            except OSError:
                pass
