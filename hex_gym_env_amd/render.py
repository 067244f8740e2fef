"""ANSI rendering of a board, same layout as HexEnv.render (HexGame.py:305-330,
HexSingleGame.py:273-298): a column header, rule lines, rows indented by 3 per row."""


def board_to_text(board, empty, black):
    """board: [N,N] array; `empty` / `black` are the encodings of EMPTY and BLACK."""
    n = board.shape[1]
    lines = [" " * 6 + "".join("  %s  |" % (j + 1) for j in range(n)),
             " " * 5 + "-" * (n * 6 - 1)]
    for i in range(n):
        cells = "".join(("  O  " if board[i, j] == empty else "  B  " if board[i, j] == black else "  W  ") + "|"
                        for j in range(n))
        lines.append(" " * (1 + i * 3) + " %s  |" % (i + 1) + cells)
        lines.append(" " * (i * 3 + 1) + "-" * (n * 7 - 1))
    return "\n".join(lines) + "\n"
