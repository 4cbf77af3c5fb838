"""The names of the reference's ``utils`` module (utils.py:21-224), bound to ``_host``'s implementations -- this file only maps
upstream's vocabulary onto them, so that code written against ``utils`` (the trainers, the drivers, user scripts) keeps working."""
from . import _host
from ._host import NullSummaryWriter, logger

get_logger = _host.package_logger
set_seed = _host.seed_host_rngs
load_config = _host.merge_saved_config
print_dict_byline = _host.log_mapping
format_metric_dict = _host.round_floats
count_parameters = _host.trainable_parameter_count
timer = _host.stopwatch

pytorch_optimizer = _host.build_optimizer
pytorch_lr_scheduler = _host.build_scheduler
reduce_lr_on_plateau = _host.plateau_step

create_flag_dict = _host.BestBook.fresh
save_checkpoint = _host.write_checkpoint
save_model_update_flag = _host.checkpoint_improved
early_stop = _host.patience_exhausted

make_summary_writer = _host.open_summary_writer
Summary = _host.ScalarSummary


def create_dir(root_dir, x):
    return _host.ensure_dir(root_dir, x)


def create_weight_dir(root_weight_dir, metrics_lst):
    return {m: _host.ensure_dir(root_weight_dir, m + '/') for m in metrics_lst}


TimeDistributed = _host.RowsAsBatch
