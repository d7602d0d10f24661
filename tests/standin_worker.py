#!/usr/bin/env python3
"""Stand-in for bin/paffy_gpu in the CPU tests of the N-GPU launcher (bin/paffy with PAFFY_GPUS=N, host/paffy_launch.c): the same
command line and environment contract -- `<cmd> [options] -i <input> -o <output>`, PAFFY_RANGE="first:end" (a stream worker reads only
its byte range), PAFFY_ROWS_FILE (a tile worker leaves the input record of every output line there, uint32 each), exit status as the
reference would end -- with the CPU oracle doing the work. Test infrastructure only."""
import os
import signal
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_lib as O  # noqa: E402


def tag(line, name, default):
    i = line.find(b"\t" + name + b":i:")
    return default if i < 0 else int(line[i + 6:].split(b"\t")[0])


def main():
    args = sys.argv[1:]
    cmd = args[0]
    # the launcher puts its own `-i <input> -o <output>` behind the user's options (one of which may have "-i" as its VALUE: `-l -i`)
    at = len(args) - 1 - args[::-1].index("-o")
    assert args[at - 2] == "-i", args
    inp, out = args[at - 1], args[at + 1]
    with open(inp, "rb") as fh:
        data = fh.read()
    rg = os.environ.get("PAFFY_RANGE")
    if rg:
        a, b = (int(x) for x in rg.split(":"))
        data = data[a:b]
    if cmd == "tile":
        res, err = O.tile(data)
        if not err.code and res and os.environ.get("PAFFY_ROWS_FILE"):  # like the worker of round 3: no output, no list (the launcher must cope)
            lines = data.splitlines(keepends=True)
            order = sorted(range(len(lines)), key=lambda k: (-tag(lines[k], b"s1", -1), -tag(lines[k], b"AS", 0), k))
            with open(os.environ["PAFFY_ROWS_FILE"], "wb") as fh:
                fh.write(struct.pack(f"<{len(order)}I", *order))
        if err.code:
            res = b""  # tile writes after the last record
    else:
        kinds = {"invert": [O.INVERT], "shatter": [O.SHATTER], "trim": [O.TRIM_IDENTITY]}[cmd]
        res, err = O.run([O.stage(k) for k in kinds], data)
    with open(out, "wb") as fh:
        fh.write(res)
    if err.code:
        status = O.lib().po_error_exit_status(err.code)
        if status == 134:
            os.kill(os.getpid(), signal.SIGABRT)
        sys.exit(status or 1)


if __name__ == "__main__":
    main()
