"""Replay of calls recorded from a run of the REFERENCE'S OWN FullSystem (oracle/ref_py.System): every trackNewCoarse and every optimize
the reference executed on a synthetic sequence is re-run from its recorded inputs through another implementation (the CPU oracle or the
HIP library) and compared with what the reference produced.  Test infrastructure only."""
import numpy as np

SCALE_A, SCALE_B = 10.0, 1000.0


def make_sequence(synth, w, h, n_frames, step=1.0, seed=None):
    """A smooth camera path in front of the plane world (sideways + forward drift, gentle rotation)."""
    K4 = synth.default_intrinsics(w, h)
    world = synth.PlaneWorld(synth.SEED if seed is None else seed)
    imgs, poses = [], []
    for k in range(n_frames):
        s = step * k
        xi = np.array([0.02 * s, 0.004 * s, 0.006 * s, 0.0015 * s, -0.003 * s, 0.001 * s])
        Rm, t = synth.se3_exp(xi)
        img, _ = world.render(K4, Rm, t, w, h)
        imgs.append(img); poses.append(synth.pose7(Rm, t))
    return K4, imgs, poses


def run_reference(R, synth, w, h, n_frames, step=1.0, point_density=600, max_frames=7, seed=None):
    K4, imgs, poses = make_sequence(synth, w, h, n_frames, step, seed)
    S = R.System(w, h, K4, point_density=point_density, max_frames=max_frames)
    status = [S.add_frame(img) for img in imgs]
    return dict(K4=K4, imgs=imgs, poses_true=poses, status=status, events=S.events(), trajectory=S.trajectory(), system=S)


def window_case(ev, imgs, w, h):
    """An opt_in event as the case dictionary oracle_py.BAWindow / dmvio_amd.BundleAdjusterHip.set_case understand."""
    F = ev["F"]
    fr = ev["frames"]
    aff_zero = np.array([[f["state_zero"][6] * SCALE_A, f["state_zero"][7] * SCALE_B] for f in fr])
    return dict(K4=ev["calib"], w=w, h=h, n_frames=F, imgs=[imgs[f["shell_id"]] for f in fr], poses0=[f["evalPT"] for f in fr], aff=aff_zero,
                exposure=np.array([f["exposure"] for f in fr]), frameIDs=np.array([f["frameID"] for f in fr]), host=ev["host"], u=ev["u"], v=ev["v"],
                idepth0=ev["idepth"], color=ev["color"], weights=ev["weights"], hasDepthPrior=ev["hasDepthPrior"], res_point=ev["res_point"], res_target=ev["res_target"])


def apply_window_state(W, ev):
    """state / state_zero / thresholds / calibration zero / marginalisation prior of the recorded window on top of a fresh one."""
    for k, f in enumerate(ev["frames"]):
        W.set_frame_zero(k, f["state_zero"])
        W.set_frame_state(k, f["state"])
    W.set_frame_energy_th(np.array([f["frameEnergyTH"] for f in ev["frames"]], np.float32))
    W.set_calib_values(ev["calib_value"], ev["calib_zero"])
    W.set_marg_prior(ev["HM"], ev["bM"])


def pair_events(events):
    """[(setref, track_in, track_out)], [(opt_in, opt_out)] in call order."""
    tracks, opts = [], []
    last_ref = {}
    pending_t = pending_o = None
    for e in events:
        if e["kind"] == "setref":
            last_ref[e["ref_id"]] = e
        elif e["kind"] == "track_in":
            pending_t = e
        elif e["kind"] == "track_out":
            tracks.append((last_ref.get(e["ref_id"]), pending_t, e)); pending_t = None
        elif e["kind"] == "opt_in":
            pending_o = e
        elif e["kind"] == "opt_out":
            opts.append((pending_o, e)); pending_o = None
    return tracks, opts


KINDS = ["setref", "track_in", "track_out", "opt_in", "opt_out"]


def load_golden(path):
    """The events of tests/golden/reference_run_*.npz (made by tests/golden/make_reference_run.py) as run_reference returns them."""
    z = np.load(path)
    n = int(z["n_events"][0])
    events = [dict() for _ in range(n)]
    frames = {}
    for key in z.files:
        if not key.startswith("e"):
            continue
        parts = key.split("__")
        k = int(parts[0][1:])
        if parts[1] == "kind":
            events[k]["kind"] = KINDS[int(z[key][0])]
        elif parts[1] == "frames":
            frames.setdefault(k, {}).setdefault(int(parts[2]), {})[parts[3]] = z[key]
        else:
            v = z[key]
            events[k][parts[1]] = v if v.ndim else v.item()
    for k, fr in frames.items():
        lst = []
        for fk in sorted(fr):
            f = {a: (b if b.ndim else b.item()) for a, b in fr[fk].items()}
            lst.append(f)
        events[k]["frames"] = lst
    meta = z["meta"]
    return dict(w=int(meta[0]), h=int(meta[1]), n_frames=int(meta[2]), density=int(meta[3]), step=float(z["step"][0]), events=events,
                trajectory={k[6:]: z[k] for k in z.files if k.startswith("traj__")})
