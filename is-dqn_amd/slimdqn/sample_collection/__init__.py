from typing import NewType

ReplayItemID = NewType("ReplayItemID", int)
