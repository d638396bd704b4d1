"""Training / testing driver with the behaviour of ``ctc_fast/runNNet.py``:
the same flags (``--layerSize --numLayers --temporalLayer --momentum --epochs --step --anneal
--reg --dataDir --alisDir --startFile --numFiles --inputDim --rawDim --outputDim --maxUttLen
--save_every --cfg_file --test``), the same on-disk run directory:

    cfg.json     the options                         (runNNet.py:71-74,139-141; run_utils.py:10-16)
    params.pk    two consecutive pickles: SGD state, then the NNet stack   (runNNet.py:181-185)
    params.pk.epochNN                                 (runNNet.py:199-201)
    epoch, num_files, last_cost, sentinel, train.log  (runNNet.py:143-205)

and the same control flow: seeds 33, shard permutation per epoch, one-deep shard prefetch,
resume from ``epoch``/``num_files``/``params.pk`` with the learning rate re-derived as
``step / anneal**start_epoch``, learning rate annealed after every epoch.

Site-specific pieces of the reference (``run_cfg.py`` paths, git revision, the SCAIL output
directory of test mode) are replaced by ``--outputDir`` / ``--likDir``.  ``--minibatch`` and
torch.distributed data parallelism are the extensions of sgd.py.
"""
import argparse
import json
import logging
import os
import pickle
import random
import socket
import time

import numpy as np

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only: RCCL needs it set early

import cudamat as cm  # noqa: E402
import dataLoader as dl
import nnets.brnnet as rnnet
import sgd
from writeLikelihoods import writeLogLikes


def build_parser():
    p = argparse.ArgumentParser(description="CTC BRNN trainer on MI355X (runNNet.py surface)")
    p.add_argument("--cfg_file", default=None, help="cfg.json of a previous run (resume / test)")
    p.add_argument("--test", action="store_true")
    p.add_argument("--outputDir", default=None, help="run directory (default: runs/<time>)")
    p.add_argument("--likDir", default=None, help="test mode: where the log-likelihoods go")
    # architecture
    p.add_argument("--layerSize", type=int, default=1824)
    p.add_argument("--numLayers", type=int, default=5)
    p.add_argument("--temporalLayer", type=int, default=3)
    # optimisation
    p.add_argument("--momentum", type=float, default=0.95)
    p.add_argument("--epochs", type=int, default=20)
    p.add_argument("--step", type=float, default=1e-5)
    p.add_argument("--anneal", type=float, default=1.3)
    p.add_argument("--reg", type=float, default=0.0)
    p.add_argument("--minibatch", type=int, default=1, help="utterances per step (reference: 1)")
    # data
    p.add_argument("--dataDir", default="./")
    p.add_argument("--alisDir", default=None)
    p.add_argument("--startFile", type=int, default=1)
    p.add_argument("--numFiles", type=int, default=384)
    p.add_argument("--inputDim", type=int, default=41 * 15)
    p.add_argument("--rawDim", type=int, default=41 * 15)
    p.add_argument("--outputDim", type=int, default=35)
    p.add_argument("--maxUttLen", type=int, default=1500)
    # save / load
    p.add_argument("--save_every", type=int, default=10)
    p.add_argument("--run_desc", default="")
    return p


def _write(path, text):
    """small bookkeeping files are replaced atomically: a reader never sees an empty file"""
    tmp = "%s.tmp%d" % (path, os.getpid())
    with open(tmp, "w") as f:
        f.write(text)
    os.replace(tmp, path)


def _bcast(obj, world):
    """rank 0's value on every rank (run directory, resume position)"""
    if world <= 1:
        return obj
    import torch.distributed as dist
    box = [obj]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def run(args=None):
    opts = build_parser().parse_args(args)
    # data-parallel extension: under `python -m torch.distributed.run --nproc-per-node N
    # runNNet.py ... --minibatch M` every rank loads the same shards with the same seeds, processes
    # its share of each minibatch and all-reduces the gradients (sgd.py / dist_sgd.py); rank 0 owns
    # the run directory and the resume bookkeeping and broadcasts both
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            import torch
            # the device must be chosen before the first (object) collective
            dev = int(os.environ.get("CUDA_DEVICE", os.environ.get("LOCAL_RANK", "0")))
            torch.cuda.set_device(dev % max(1, torch.cuda.device_count()))
            dist.init_process_group(os.environ.get("SCTC_DIST_BACKEND", "nccl"), rank=rank,
                                    world_size=world)
    if opts.cfg_file:
        with open(opts.cfg_file) as f:
            cfg = json.load(f)
        for k in ("test", "startFile", "likDir"):
            cfg[k] = getattr(opts, k)
        if opts.test:                      # runNNet.py:96-99: test data comes from the command line
            cfg["dataDir"], cfg["numFiles"] = opts.dataDir, opts.numFiles
            cfg["alisDir"] = opts.alisDir
        output_dir = cfg["output_dir"]
    else:
        cfg = vars(opts).copy()
        # every rank must arrive at the same (fresh, timestamped) directory: rank 0 names it
        output_dir = _bcast(opts.outputDir or os.path.join("runs", time.strftime("%Y%m%d_%H%M%S")),
                            world)
        os.makedirs(output_dir, exist_ok=True)
        cfg["cfg_file"] = os.path.join(output_dir, "cfg.json")
    cfg["output_dir"] = output_dir
    cfg["in_file"] = cfg["out_file"] = os.path.join(output_dir, "params.pk")
    cfg["host"], cfg["pid"] = socket.gethostname(), os.getpid()
    cfg.setdefault("reg", 0.0)
    cfg.setdefault("minibatch", 1)
    o = argparse.Namespace(**cfg)

    master = rank == 0
    o.master = master
    logging.basicConfig(filename=os.path.join(output_dir, ("test.log" if o.test else "train.log") +
                                              ("" if master else ".rank%d" % rank)),
                        level=logging.DEBUG, force=True)
    logger = logging.getLogger()
    if master:
        logger.addHandler(logging.StreamHandler())
    logger.info("Running on %s" % o.host)
    np.random.seed(33)                     # runNNet.py:112-115
    random.seed(33)
    import torch
    dev = int(os.environ.get("CUDA_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    cm.cuda_set_device(dev % max(1, torch.cuda.device_count()))
    if o.test:
        return test(o, logger)

    loader = dl.DataLoader(o.dataDir, o.rawDim, o.inputDim, o.alisDir or o.dataDir)
    nn = rnnet.NNet(o.inputDim, o.outputDim, o.layerSize, o.numLayers, o.maxUttLen,
                    temporalLayer=o.temporalLayer, reg=o.reg, maxUtts=max(1, o.minibatch))
    nn.initParams()
    opt = sgd.SGD(nn, o.maxUttLen, alpha=o.step, momentum=o.momentum, minibatch=o.minibatch)
    cfg["param_count"] = int(nn._param_count)
    if master:
        with open(cfg["cfg_file"], "w") as f:
            json.dump(cfg, f, indent=1, sort_keys=True)

    epoch_file = os.path.join(output_dir, "epoch")
    num_files_file = os.path.join(output_dir, "num_files")
    # only rank 0 reads the bookkeeping files (it is also their only writer)
    start_epoch, resume_file, have_ckpt = 0, None, False
    if master:
        start_epoch = int(open(epoch_file).read()) + 1 if os.path.exists(epoch_file) else 0
        if os.path.exists(num_files_file):
            resume_file = int(open(num_files_file).read().strip())
        have_ckpt = os.path.exists(o.in_file)
    start_epoch, resume_file, have_ckpt = _bcast((start_epoch, resume_file, have_ckpt), world)
    if have_ckpt:                          # resume, runNNet.py:150-154
        with open(o.in_file, "rb") as fid:
            opt.fromFile(fid)
            opt.alpha = opt.alpha / (o.anneal ** start_epoch)
            nn.fromFile(fid)

    for k in range(start_epoch, o.epochs):
        perm = np.random.permutation(o.numFiles) + 1
        file_start = 0
        if k == start_epoch and resume_file is not None:
            file_start = resume_file
            logger.info("Starting from file %d, epoch %d" % (file_start, start_epoch))
        elif master:
            _write(num_files_file, str(file_start))
        if file_start < perm.shape[0]:
            loader.loadDataFileAsynch(int(perm[file_start]))
        for i in range(file_start, perm.shape[0]):
            start = time.time()
            data_dict, alis, keys, sizes = loader.getDataAsynch()
            if i + 1 < perm.shape[0]:
                loader.loadDataFileAsynch(int(perm[i + 1]))      # prefetch
            opt.run(data_dict, alis, keys, sizes)
            logger.info("File time %f" % (time.time() - start))
            if master and (i + 1) % o.save_every == 0:
                logger.info("Saving parameters")
                with open(o.out_file, "wb") as fid:
                    opt.toFile(fid)
                    nn.toFile(fid)
                _write(num_files_file, "%d" % (i + 1))
                if opt.expcost:
                    last = opt.expcost[-1] - (opt.regcost[-1] if (o.reg > 0.0 and opt.regcost) else 0.0)
                    _write(os.path.join(output_dir, "last_cost"), str(last))
        if master:
            _write(epoch_file, str(k))
            # deliberate deviation: the reference leaves num_files at numFiles here, so a job
            # resumed right after a completed epoch skips the whole next epoch (runNNet.py:163-169)
            _write(num_files_file, "0")
            with open(o.out_file + ".epoch{0:02}".format(k), "wb") as fid:
                opt.toFile(fid)
                nn.toFile(fid)
        opt.alpha = opt.alpha / o.anneal
    if master:
        _write(os.path.join(output_dir, "sentinel"), "")   # run complete (run_utils.touch_file)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    return opt, nn


def test(o, logger):
    with open(o.in_file, "rb") as fid:
        pickle.load(fid, encoding="latin1")    # SGD data, not needed (runNNet.py:217-218); py2 pickles
        loader = dl.DataLoader(o.dataDir, o.rawDim, o.inputDim, o.alisDir or o.dataDir)
        nn = rnnet.NNet(o.inputDim, o.outputDim, o.layerSize, o.numLayers, o.maxUttLen,
                        temporalLayer=o.temporalLayer, train=False, maxUtts=16)
        nn.fromFile(fid)
    out_dir = o.likDir or os.path.join(o.output_dir, "ctc_loglikes")
    os.makedirs(out_dir, exist_ok=True)
    for i in range(o.startFile, o.numFiles + 1):
        logger.info("Running file %d" % i)
        writeLogLikes(loader, nn, i, out_dir, writePickle=True)
    return out_dir


if __name__ == "__main__":
    run()
