module github.com/lattigo-amd/hering

go 1.21

require github.com/tuneinsight/lattigo/v6 v6.2.0
