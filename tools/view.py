#!/usr/bin/env python3
"""Live viewer over HTTP: renders a workload on the GPU and serves the frames to a browser, whose mouse
drives the camera as the reference's window does (drag = rotate, wheel = zoom).

    python tools/view.py [workload] [--port 8080] [--host 127.0.0.1] [--seconds S] [--png-dir DIR]
"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", nargs="?", default="cfg2_julia_1080p")
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=8080)
    ap.add_argument("--seconds", type=float, default=None, help="stop after this long (default: until Ctrl-C)")
    ap.add_argument("--png-dir", default=None, help="also keep every presented frame as a PNG here")
    args = ap.parse_args()
    import kifs_raymarching_amd as K
    from kifs_raymarching_amd.configs import WORKLOADS
    from kifs_raymarching_amd.viewer import HttpSink, PngSequenceSink, ViewerSession
    w = WORKLOADS[args.workload]
    with K.GraphicState(0, screen_data=w.screen, camera_data=w.camera, gui_data=w.gui) as gs:
        gs.set_iters(*w.iters)
        session = ViewerSession(gs)
        http = HttpSink(args.host, args.port, on_input=session.handle)
        session.sinks.append(http)
        if args.png_dir:
            session.sinks.append(PngSequenceSink(args.png_dir))
        print(f"viewer: {http.url}  ({args.workload}, {w.screen.width}x{w.screen.height})", flush=True)
        session.run(args.seconds)
        print(f"{session.frames_presented} frames presented; last frame {session.last_frame_ms:.2f} ms "
              "(render + copy to the host)")
        session.close()


if __name__ == "__main__":
    main()
