"""Mirror of the reference's ``utils`` package for the hot path (losses, SinkhornDistance, LR schedule)."""
