"""Validation-set analysis: the second caller of the hot path (mirrors
jarvis/analysis/analyze.py:22-96).

`analyze_frames` is the reference's loop over validation frame sets -- predict every
frame set, keep the ones the network detected, write `frame_names.csv`,
`points_HybridNet.csv` and `points_GroundTruth.csv` with the same numpy.savetxt calls
(hence byte-identical files for identical predictions).  Project management and the
Dataset3D loader are outside the hot path: `analyze_validation_data`
keeps the reference's signature but takes the configuration and the dataset from the
caller instead of constructing them from a project name.
"""
import os
import time

import numpy as np
from numpy import savetxt

from ..prediction.jarvis3D import JarvisPredictor3D
from ..utils.reprojection import load_reprojection_tools


def analyze_frames(predictor, samples, reproTools, output_dir, num_joints, progress_bar=None,
                   num_frame_sets=None):
    """samples: iterable of batch-1 collated Dataset3D analysis samples
    `[imgs (1,C,H,W,3), keypoints3D (1,J,3), ..., dataset_name [str], file_name [str]]`
    (dataset3D.py:248-258 behind a DataLoader(batch_size=1), analyze.py:46-51).
    Returns (number of frame sets seen, number predicted)."""
    pointsNet, pointsGT, filenames = [], [], []
    seen = 0
    for item, sample in enumerate(samples):
        seen += 1
        if progress_bar is not None and num_frame_sets:
            progress_bar.progress(float(item + 1) / num_frame_sets)
        keypoints3D = sample[1][0].numpy()
        imgs_orig = sample[0][0]
        dataset_name = sample[-2][0]
        reproTool = reproTools[dataset_name]
        file_name = sample[-1][0]
        imgs = imgs_orig.cuda().float().permute(0, 3, 1, 2)          # analyze.py:66
        points3D_net, _ = predictor(imgs.contiguous(), reproTool.cameraMatrices.cuda(),
                                    reproTool.intrinsicMatrices.cuda(),
                                    reproTool.distortionCoefficients.cuda())
        if points3D_net is not None:
            pointsNet.append(points3D_net[0].cpu().detach().numpy())
            pointsGT.append(keypoints3D)
            filenames.append(file_name)
    os.makedirs(output_dir, exist_ok=True)
    savetxt(os.path.join(output_dir, "frame_names.csv"), np.array(filenames), delimiter=",", fmt="%s")
    savetxt(os.path.join(output_dir, "points_HybridNet.csv"),
            np.array(pointsNet).reshape((-1, num_joints * 3)), delimiter=",")
    savetxt(os.path.join(output_dir, "points_GroundTruth.csv"),
            np.array(pointsGT).reshape((-1, num_joints * 3)), delimiter=",")
    return seen, len(pointsNet)


def analyze_validation_data(project_name=None, weights_center="latest", weights_hybridnet="latest",
                            cameras_to_use=None, progress_bar=None, *, cfg=None, dataset=None,
                            output_root=None, reproTools=None):
    """analyze.py:22-96 with the project manager and Dataset3D supplied by the caller:
    cfg = the project's configuration, dataset = a Dataset3D(cfg, set='val',
    analysisMode=True)-shaped sequence.  Returns the output directory."""
    if cfg is None or dataset is None:
        raise NotImplementedError(
            "project management and Dataset3D are outside this package (SURVEY section 2): pass "
            "cfg= and dataset= (project %r)" % (project_name,))
    from torch.utils.data import DataLoader
    root = output_root if output_root is not None else os.path.join(
        cfg.PARENT_DIR, getattr(cfg, "PROJECTS_ROOT_PATH", "projects"), str(project_name), "analysis")
    output_dir = os.path.join(root, "Validation_Predictions_" + time.strftime("%Y%m%d-%H%M%S"))
    os.makedirs(output_dir)
    predictor = JarvisPredictor3D(cfg, weights_center, weights_hybridnet)
    if reproTools is None:
        reproTools = load_reprojection_tools(cfg, cameras_to_use=cameras_to_use)
    loader = DataLoader(dataset, batch_size=1, shuffle=False,
                        num_workers=getattr(cfg, "DATALOADER_NUM_WORKERS", 0), pin_memory=True)
    seen, done = analyze_frames(predictor, loader, reproTools, output_dir,
                                cfg.KEYPOINTDETECT.NUM_JOINTS, progress_bar, len(dataset))
    if done != seen:
        print("Network could not detect instance in %d frameSets. Those were not included in the "
              "output files!" % (seen - done))
    return output_dir
