"""Mirror of src/datasets/simclr_dataset.py:3-13: two augmented views of each item."""
from torch.utils.data import Dataset


class SimCLRDataset(Dataset):
    def __init__(self, base_dataset, transform):
        self.base_dataset = base_dataset
        self.transform = transform

    def __len__(self):
        return len(self.base_dataset)

    def __getitem__(self, idx):
        img = self.base_dataset[idx][0]
        return self.transform(img), self.transform(img)
