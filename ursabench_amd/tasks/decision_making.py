"""Decision — drop-in for URSABench/tasks/decision_making.py:12-152: expected-cost decisions
`risk += p_smooth @ cost` (:129), D = argmin(risk/S), True_Cost = sum cost[y, D] (:146-151).

The reference picks the cost matrix by the *class identity* of a torchvision dataset (:90-97).
torchvision is not a dependency here: pass `cost_mat=` explicitly, or let the fallback match the
dataset class NAME ('MNIST', 'CIFAR10', 'CIFAR100'); anything else is NotImplementedError, as in
the reference."""
import torch

from .task_base import EnsembleAccumulator, _Task, as_member_list

__all__ = ['Decision', 'MNIST_cost', 'CIFAR10_cost', 'CIFAR100_cost', 'decision_cost']

# CIFAR-100 fine-label order (decision_making.py:29-35, there named `coarse_label`)
_CIFAR100 = ('apple aquarium_fish baby bear beaver bed bee beetle bicycle bottle bowl boy bridge bus butterfly camel '
             'can castle caterpillar cattle chair chimpanzee clock cloud cockroach couch crab crocodile cup dinosaur '
             'dolphin elephant flatfish forest fox girl hamster house kangaroo computer_keyboard lamp lawn_mower '
             'leopard lion lizard lobster man maple_tree motorcycle mountain mouse mushroom oak_tree orange orchid '
             'otter palm_tree pear pickup_truck pine_tree plain plate poppy porcupine possum rabbit raccoon ray road '
             'rocket rose sea seal shark shrew skunk skyscraper snail snake spider squirrel streetcar sunflower '
             'sweet_pepper table tank telephone television tiger tractor train trout tulip turtle wardrobe whale '
             'willow_tree wolf woman worm').split()


def _cost(num_classes, important_rows, high):
    """off-diagonal 0.1, rows of the important true classes `high`, diagonal 0 (:12-50)."""
    c = torch.full((num_classes, num_classes), 0.1)
    c[important_rows] = high
    c.fill_diagonal_(0)
    return c


def MNIST_cost(num_classes):
    return _cost(num_classes, [3, 7], 100.0)


def CIFAR10_cost(num_classes):
    return _cost(num_classes, [0, 1, 8, 9], 1.0)          # plane, automobile, ship, truck


def CIFAR100_cost(num_classes):
    rows = [i for i, name in enumerate(_CIFAR100) if name in ('tank', 'rocket', 'pickup_truck')]
    return _cost(num_classes, rows, 1.0)


def decision_cost(D, y_true, cost_mat=None):
    return cost_mat[y_true, D].sum()


class Decision(_Task):
    def __init__(self, dataloader, num_classes, device, *, cost_mat=None, kernels=None, process_group=None,
                 acc_kw=None):
        super().__init__(dataloader, num_classes, device)
        self.data_loader = dataloader['decision_data_test']
        self.num_classes = num_classes
        self.device = device
        self.process_group = process_group
        self.targets = torch.cat([y.cpu() for _, y in self.data_loader])
        if cost_mat is None:
            name = type(self.data_loader.dataset).__name__
            maker = {'MNIST': MNIST_cost, 'CIFAR10': CIFAR10_cost, 'CIFAR100': CIFAR100_cost}.get(name)
            if maker is None:
                raise NotImplementedError
            cost_mat = maker(self.num_classes)
        self.cost_mat = cost_mat.float().cpu()
        self._acc = EnsembleAccumulator(self.data_loader, num_classes, device, kernels, smoothed=True,
                                        with_entropy=False, cost=self.cost_mat.to(device).contiguous(), **(acc_kw or {}))
        self.reset()

    def _publish(self):
        p, _, r, n = self._acc.reduced(self._local_count, self.process_group)   # update_statistics only
        self.ensemble_proba, self.risk, self.num_samples_collected = p, r, n

    def reset(self):
        self._local_count = 0
        self.num_samples_collected = 0
        self._acc.reset()
        self.ensemble_proba, _, self.risk = self._acc.local()

    def update_statistics(self, models, output_performance=True, smoothing=True):
        members = as_member_list(models)
        self._local_count += len(members)
        self._acc.accumulate(members)
        self._publish()
        if output_performance:
            return self.get_performance_metrics(output_performance, smoothing)

    def get_performance_metrics(self, output_performance=False, smoothing=True):
        D = (self.risk / self.num_samples_collected).argmin(1)
        return {'True_Cost': decision_cost(D, self.targets, self.cost_mat), 'Decision': D, 'Pred_cost': self.risk}
