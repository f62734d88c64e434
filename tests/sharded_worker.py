"""Worker of the multi-process rehearsal in test_gpu_sharded.py: one rank of a world of N processes
that all use cuda:0 (a one-GPU box), talking over gloo.  The code under test is exactly what an
8-GPU run executes -- gp_dla_detection_amd.distributed.process_qsos_sharded and its multi-DLA
sibling -- only the backend (gloo for RCCL) and the device mapping (every rank on cuda:0) differ.

The processes are forked from a fork server that conftest.py starts BEFORE anything touches the
GPU, so no process that holds a HIP context ever forks or execs.
"""
import os

import numpy as np


def build_case(kind):
    """Seeded inputs, identical in every process."""
    import gp_dla_detection_amd as gp
    from gp_dla_detection_amd import synthetic
    from gp_dla_detection_amd.parameters import MultiParameters
    model = synthetic.make_model(20)
    sizes = [310, 640, 222, 801, 415, 333, 560]
    spectra = [synthetic.make_spectrum(830 + i, n, model, mask_fraction=0.04 if i % 2 else 0.0)
               for i, n in enumerate(sizes)]
    cat = synthetic.make_prior_catalog()
    z = np.array([s["z_qso"] for s in spectra])
    if kind == "single":
        samples = synthetic.make_samples(700)
        return model, samples, spectra, gp.dla_existence_prior(cat["z_qsos"], cat["dla_ind"], z), None
    p = MultiParameters(max_dlas=3)
    samples = synthetic.make_samples(384)
    lp = gp.dla_existence_prior_multi(cat["z_qsos"], cat["dla_ind"], z, 0.31, 0.69, p)
    return model, samples, spectra, lp, p


def run_rank(rank, world, port, kind, out_dir, num_quasars):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    from gp_dla_detection_amd import distributed
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model, samples, spectra, lp, p = build_case(kind)
        spectra = spectra[:num_quasars]
        lp = tuple(np.asarray(x)[:num_quasars] for x in lp)
        counts = [np.asarray(s["wavelengths"]).size for s in spectra]
        loaded = []

        def loader(lo, hi):  # a rank reads ONLY its block
            loaded.append((lo, hi))
            return spectra[lo:hi]

        if kind == "single":
            fields, block, local = distributed.process_qsos_sharded(
                model, samples, loader, lp, device=0, pixel_counts=counts, max_quasars_per_batch=2)
            local = {} if local is None else {"sample_log_likelihoods_dla": local}
        else:
            fields, block, local = distributed.process_qsos_multiple_dlas_meanflux_sharded(
                model, samples, loader, lp, params=p, device=0, pixel_counts=counts, max_quasars_per_batch=2)
            local = local or {}
        assert loaded == ([block] if block[1] > block[0] else [])
        np.savez(os.path.join(out_dir, f"{kind}_w{world}_r{rank}.npz"), block=np.array(block),
                 **{"f_" + k: v for k, v in fields.items()}, **{"l_" + k: v for k, v in local.items()})
    finally:
        if world > 1:
            dist.destroy_process_group()


def run_rccl_world1(port, out_dir):
    """RCCL itself cannot be given two ranks on one GPU (it refuses duplicate devices), so on a
    one-GPU box the RCCL leg is exercised at world size 1: backend "nccl" is initialised and its
    all-gather runs on the library-owned summary table exactly as bench.py / distributed.py hand
    it over -- a zero-copy torch view of HBM that libgpdla.so allocated."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    import torch
    import torch.distributed as dist

    import gp_dla_detection_amd as gp
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        model, samples, spectra, lp, _ = build_case("single")
        stream = torch.cuda.Stream()
        ctx = gp.Context(0, stream=stream)
        ctx.set_model(model)
        ctx.set_samples(samples)
        batch = ctx.upload(spectra, lp[0], lp[1])
        with torch.cuda.stream(stream):
            batch.process()
            table = batch.summary_tensor()
            out = torch.empty_like(table)
            dist.all_gather_into_tensor(out, table)       # RCCL reads the library's buffer in place
            total = table[:, 5].clone()
            dist.all_reduce(total, op=dist.ReduceOp.MAX)
        torch.cuda.synchronize()
        ref = batch.download()
        np.savez(os.path.join(out_dir, "rccl_w1.npz"), gathered=out.cpu().numpy(), table=table.cpu().numpy(),
                 ll=ref["log_likelihoods_dla"], backend=np.array(dist.get_backend()))
        batch.close()
        ctx.close()
    finally:
        dist.destroy_process_group()


def run_files_rank(rank, world, port, multi, in_dir, out_dir, per_batch, fail_rank=None):
    """One rank of the file-to-file run (gp_dla_detection_amd.run_dr12q.run) on cuda:0 over gloo:
    reads the synthetic file set in ``in_dir`` (synthetic.write_file_set), writes its chunk file
    into ``out_dir`` and saves what it returned.  ``fail_rank``: that rank's reader raises when it is
    asked for its second batch (a damaged input file); every rank then records how the run ended."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import datetime
    import time

    import torch
    import torch.distributed as dist

    from gp_dla_detection_amd import io, run_dr12q
    from gp_dla_detection_amd.parameters import MultiParameters
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    if fail_rank is not None and rank == fail_rank:
        real, calls = io.PreloadedReader.read_csr, []

        def read_csr(self, indices, z_qsos):
            calls.append(len(indices))
            if len(calls) == 2:
                raise OSError("injected: cell 2 of preloaded_qsos.mat cannot be read")
            return real(self, indices, z_qsos)
        io.PreloadedReader.read_csr = read_csr
    t0 = time.perf_counter()
    try:
        pr = np.load(os.path.join(in_dir, "prior_catalog.npz"))
        res = run_dr12q.run(os.path.join(in_dir, "preloaded_qsos.mat"), os.path.join(in_dir, "catalog.mat"),
                            os.path.join(in_dir, "learned_qso_model_synthetic.mat"),
                            os.path.join(in_dir, "dla_samples.mat"), out_dir, "synth",
                            prior_catalog=dict(z_qsos=pr["z_qsos"], dla_ind=pr["dla_ind"]), multi=multi,
                            params=MultiParameters(max_dlas=3) if multi else None, Z_lls=0.31, Z_dla=0.69,
                            device=0, max_quasars_per_batch=per_batch,
                            run_metadata=dict(release="dr12q", training_release="dr12q",
                                              training_set_name="synthetic"))
        np.savez(os.path.join(out_dir, f"files_{'multi' if multi else 'single'}_w{world}_r{rank}.npz"),
                 block=np.array(res["block"]), chunk=np.array(res["chunk"] or ""), selected=res["selected"],
                 **{"f_" + k: v for k, v in res["fields"].items()})
    except Exception as e:
        if fail_rank is None:
            raise
        with open(os.path.join(out_dir, f"ended_r{rank}.txt"), "w") as f:
            f.write(f"{type(e).__name__}\n{time.perf_counter() - t0}\n")
    finally:
        if world > 1:
            dist.destroy_process_group()


def run_cli(argv):
    """`python -m gp_dla_detection_amd.run_dr12q <argv>` in a clean child of the fork server."""
    os.environ.update(MASTER_ADDR="127.0.0.1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    from gp_dla_detection_amd import run_dr12q
    run_dr12q.main(list(argv))
