# TEST INFRASTRUCTURE ONLY.  Prints whole top-level function definitions of a reference source file, selected by the exact text
# their first line starts with (after indentation): from that line to the next line that is just "}" at the same indentation.  Used by oracle/Makefile to hand single member
# functions of src/Frame.cc / src/KeyFrame.cc / src/MapPoint.cc (files that cannot be compiled whole here: they pull in g2o,
# Eigen, calib3d ...) to the compiler through a temporary file that is deleted after the build; nothing is copied into the repo.
#   awk -v sigs='void Frame::AssignFeaturesToGrid()|bool Frame::PosInGrid(' -f ref_excerpt.awk /root/reference/src/Frame.cc
BEGIN { n = split(sigs, S, "|"); found = 0 }
{
    if (!on) {
        line = $0
        sub(/^[ \t]+/, "", line)
        for (i = 1; i <= n; i++)
            if (index(line, S[i]) == 1) { on = 1; found++; indent = substr($0, 1, length($0) - length(line)) }
    }
    if (on) {
        print
        if ($0 == indent "}") on = 0
    }
}
END { if (found != n) { print "ref_excerpt.awk: matched " found " of " n " signatures" > "/dev/stderr"; exit 1 } }
