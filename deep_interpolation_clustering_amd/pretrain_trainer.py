"""Drop-in for the reference's ``pretrain_trainer`` module: ``Trainer`` for the interpolation auto-encoder
(pretrain_trainer.py:17-438).  The epoch machinery lives in ``_trainer_common.TrainerBase``."""
from ._trainer_common import TrainerBase
from .utils import format_metric_dict, logger, timer


class Trainer(TrainerBase):
    restore_attr = 'restore_metric'

    def __init__(self, args, model, dl_dict, exp_path, device, **kwags):
        super().__init__(args, model, dl_dict, exp_path, device, **kwags)

    def train(self):
        """pretrain_trainer.py:64-88: epochs of train -> validate -> checkpoint-on-improvement -> early stop."""
        logger.info('*******Building the model*******')
        if self.args.restore:
            self.load_weight()
        with timer('Duration of training'):
            for epoch in range(1, self.args.max_epochs):
                train_metrics = self.train_one_epoch(self.train_dl, denoise=self.args.denoise)
                logger.info('==> Epoch: {}, Train, {}'.format(epoch, format_metric_dict(train_metrics)))
                valid_metrics, _ = self.eval_one_epoch('valid', self._eval_dl('validation'), denoise=self.args.denoise)
                # (upstream formats the dict for a debug line here, pretrain_trainer.py:77 -- which ROUNDS it in place to 4 decimals: the
                #  best-checkpoint comparisons of aly_pred see the rounded values, so ties between epochs resolve as they do upstream)
                logger.debug('{}'.format(format_metric_dict(valid_metrics)))
                verdict = self.aly_pred('valid', valid_metrics)
                self.epoch += 1
                if verdict['early_stop']:
                    logger.info('========Best model=========')
                    logger.info('{}'.format(self.flag_dict))
                    break
