"""Host-side readers/writers of the on-disk contract around the hot path (SURVEY.md
Appendix A): PFM, camera files, pair.txt, the DTU evaluation and training samples."""
from .data_io import read_pfm, save_pfm, save_pfm_rows_bottom_up
from .dtu_eval import MVSDataset, read_cam_file, read_pair_file

__all__ = ["read_pfm", "save_pfm", "save_pfm_rows_bottom_up", "MVSDataset", "read_cam_file", "read_pair_file"]


def find_dataset_def(name):
    """Reference surface (MVSNet/datasets/__init__.py): dataset class by module name."""
    if name in ("dtu_yao_eval", "dtu_eval"):
        return MVSDataset
    if name in ("dtu_yao", "dtu_train"):
        from .dtu_train import MVSDataset as TrainDataset
        return TrainDataset
    raise KeyError(f"dataset {name!r} is not part of this build (DTU evaluation and training loaders only)")
